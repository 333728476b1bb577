"""Lazy quantity tree over sample chunks [M, n, 2] (reference interface: mlmc/quantity/quantity.py:14-695).

Host-side caller of the hot path: it selects / combines stored samples chunk by chunk and hands raw
[M, n, 2] blocks to the device estimators (quantity_estimate.py).  Same public names and conventions as
the reference (`make_root_quantity`, `Quantity`, `QuantityConst`, `QuantityMean`, `QuantityStorage`), own
implementation.  Element-wise algebra of the tree runs in NumPy here (SURVEY 8(f) row 1 moves it to the
device); moment functions, NaN masking and all level sums run on the GPU.
"""
import operator

import numpy as np

from . import quantity_types as qt
from .quantity_spec import ChunkSpec

RNG = np.random.default_rng()

# sample-chunk memo: every node keeps the chunks it evaluated until the next cache_clear()
# (the reference memoises Quantity.samples per (level, chunk, size, id), quantity.py:117-135)
_cache_generation = [0]


def cache_clear():
    _cache_generation[0] += 1


def make_root_quantity(storage, q_specs):
    """Root quantity of a storage: dict of time series of fields of arrays (reference: quantity.py:14-32)."""
    entries = []
    for spec in q_specs:
        array_type = qt.ArrayType(spec.shape, qt.ScalarType(float))
        field_type = qt.FieldType([(loc, array_type) for loc in spec.locations])
        entries.append((spec.name, qt.TimeSeriesType(spec.times, field_type)))
    return QuantityStorage(storage, qt.DictType(entries))


class Quantity:
    def __init__(self, quantity_type, operation, input_quantities=[]):
        self.qtype = quantity_type
        self._operation = operation
        self._input_quantities = list(input_quantities)
        self._storage = self.get_quantity_storage()
        self._selection_id = self.set_selection_id()
        self._check_selection_ids()
        self._memo = (None, {})
        # a quantity whose chunks are not a pure function of the stored samples (random sub-sampling below it)
        self._volatile = any(getattr(q, "_volatile", False) for q in self._input_quantities)
        # symbolic description of the node for the device lowering (quantity/lowering.py); None: host evaluation only
        self._sym = None

    # ---- structure ---------------------------------------------------------------------------
    def get_quantity_storage(self):
        for q in self._input_quantities:
            storage = q.get_quantity_storage()
            if storage is not None:
                self._storage = storage
                return storage
        return None

    def set_selection_id(self):
        selection_id = None
        for q in self._input_quantities:
            sid = q.selection_id()
            if selection_id is None:
                selection_id = sid
            elif sid is not None and selection_id != sid:
                raise Exception("Different selection IDs among input quantities")
        return selection_id

    def _check_selection_ids(self):
        if self._storage is None:
            return
        for q in self._input_quantities:
            sid = q.selection_id()
            if sid is not None and sid != self.selection_id():
                raise AssertionError("Not all input quantities come from the same quantity storage")

    def selection_id(self):
        if self._selection_id is not None:
            return self._selection_id
        if self._storage is None:
            self._storage = self.get_quantity_storage()
        return id(self._storage)

    def size(self) -> int:
        return self.qtype.size()

    def get_cache_key(self, chunk_spec):
        chunk_size = None
        if chunk_spec.chunk_slice is not None:
            chunk_size = chunk_spec.chunk_slice.stop - chunk_spec.chunk_slice.start
        return (chunk_spec.level_id, chunk_spec.chunk_id, chunk_size, id(self))

    def _memoised(self, chunk_spec, compute):
        gen, table = getattr(self, "_memo", (None, {}))
        if gen != _cache_generation[0]:
            table = {}
            self._memo = (_cache_generation[0], table)
        key = self.get_cache_key(chunk_spec)
        if key not in table:
            table[key] = compute()
        return table[key]

    def samples(self, chunk_spec):
        """Chunk of this quantity: ndarray [M, n, 2] (level 0: [M, n, 1])."""
        return self._memoised(chunk_spec, lambda: self._operation(*[q.samples(chunk_spec) for q in self._input_quantities]))

    # ---- selection -----------------------------------------------------------------------------
    def select(self, *args):
        masks = args[0]
        for q in args:
            if not isinstance(q.qtype.base_qtype(), qt.BoolType):
                raise Exception("Quantity: {} doesn't have BoolType, instead it has QType: {}".format(q, q.qtype.base_qtype()))
        for m in args[1:]:
            masks = np.logical_and(masks, m)

        def pick(x, mask):
            return x[..., mask, :]
        selected = Quantity(quantity_type=self.qtype, input_quantities=[self, masks], operation=pick)
        selected._selection_id = id(selected)
        selected._sym = ("select",)
        return selected

    def __array_ufunc__(self, ufunc, method, *args, **kwargs):
        return Quantity._method(ufunc, method, *args, **kwargs)

    # ---- arithmetic ----------------------------------------------------------------------------
    def __add__(self, other):
        return Quantity.create_quantity([self, Quantity.wrap(other)], Quantity.add_op)

    def __sub__(self, other):
        return Quantity.create_quantity([self, Quantity.wrap(other)], Quantity.sub_op)

    def __mul__(self, other):
        return Quantity.create_quantity([self, Quantity.wrap(other)], Quantity.mult_op)

    def __truediv__(self, other):
        return Quantity.create_quantity([self, Quantity.wrap(other)], Quantity.truediv_op)

    def __mod__(self, other):
        return Quantity.create_quantity([self, Quantity.wrap(other)], Quantity.mod_op)

    def __radd__(self, other):
        return Quantity.create_quantity([Quantity.wrap(other), self], Quantity.add_op)

    def __rsub__(self, other):
        return Quantity.create_quantity([Quantity.wrap(other), self], Quantity.sub_op)

    def __rmul__(self, other):
        return Quantity.create_quantity([Quantity.wrap(other), self], Quantity.mult_op)

    def __rtruediv__(self, other):
        return Quantity.create_quantity([Quantity.wrap(other), self], Quantity.truediv_op)

    def __rmod__(self, other):
        return Quantity.create_quantity([Quantity.wrap(other), self], Quantity.mod_op)

    @staticmethod
    def create_quantity(quantities, operation):
        """A Quantity if any operand depends on samples, otherwise a folded QuantityConst."""
        for q in quantities:
            if not isinstance(q, QuantityConst):
                result = Quantity(q.qtype, operation=operation, input_quantities=quantities)
                result._sym = ("binop", operation)
                return result
        return QuantityConst(quantities[0].qtype, value=operation(*[q._value for q in quantities]))

    def _reduction_op(self, quantities, operation):
        return Quantity.create_quantity(quantities, operation)

    add_op = staticmethod(operator.add)
    sub_op = staticmethod(operator.sub)
    mult_op = staticmethod(operator.mul)
    truediv_op = staticmethod(operator.truediv)
    mod_op = staticmethod(operator.mod)

    # ---- comparisons -> sample masks -------------------------------------------------------------
    @staticmethod
    def _process_mask(x, y, op):
        """A sample passes only if every value of it (all components, fine and coarse) meets the condition."""
        mask = op(x, y)
        return mask.all(axis=tuple(range(mask.ndim - 2))).all(axis=1)

    def _mask_quantity(self, other, op):
        other = Quantity.wrap(other)
        if not isinstance(self.qtype.base_qtype(), qt.ScalarType) or not isinstance(other.qtype.base_qtype(), qt.ScalarType):
            raise TypeError("Quantity has base qtype {}. Quantities with base qtype ScalarType are the only ones "
                            "that support comparison".format(self.qtype.base_qtype()))
        mask = Quantity(quantity_type=self.qtype.replace_scalar(qt.BoolType()), input_quantities=[self, other],
                        operation=lambda x, y: Quantity._process_mask(x, y, op))
        mask._sym = ("cmp", op)
        return mask

    def __lt__(self, other):
        return self._mask_quantity(other, operator.lt)

    def __le__(self, other):
        return self._mask_quantity(other, operator.le)

    def __gt__(self, other):
        return self._mask_quantity(other, operator.gt)

    def __ge__(self, other):
        return self._mask_quantity(other, operator.ge)

    def __eq__(self, other):
        return self._mask_quantity(other, operator.eq)

    def __ne__(self, other):
        return self._mask_quantity(other, operator.ne)

    __hash__ = object.__hash__

    # ---- sub-sampling (bootstrap) ------------------------------------------------------------------
    @staticmethod
    def pick_samples(chunk, subsample_params):
        """Draw this chunk's share of a k-of-n subsample: hypergeometric count, then a uniform choice
        (reference: quantity.py:308-325, selection sampling over chunks)."""
        import scipy.stats
        size = scipy.stats.hypergeom(subsample_params.n, subsample_params.k, chunk.shape[1]).rvs(size=1)
        out = RNG.choice(chunk, size=size, axis=1)
        subsample_params.k -= out.shape[1]
        subsample_params.n -= chunk.shape[1]
        return out

    def subsample(self, sample_vec):
        class SubsampleParams:
            def __init__(self, num_subsample, num_collected):
                self._orig_k = self.k = num_subsample
                self._orig_n = self.n = num_collected
                self._orig_total_n = self.total_n = num_collected

        per_level = {level: SubsampleParams(sample_vec[level], n_coll)
                     for level, n_coll in enumerate(self.get_quantity_storage().n_collected())}
        params_q = Quantity.wrap(hash(frozenset(per_level.items())))

        def adjust_value(values, level_id):
            p = per_level[level_id]
            p.k, p.n, p.total_n = p._orig_k, p._orig_n, p._orig_total_n
            return p
        params_q._adjust_value = adjust_value
        params_q._sym = None                       # per-level mutable state, not a constant
        picked = Quantity(quantity_type=self.qtype.replace_scalar(qt.BoolType()), input_quantities=[self, params_q],
                          operation=Quantity.pick_samples)
        picked._volatile = True          # a fresh random draw on every evaluation: never cached on the device
        picked._sym = ("subsample", per_level)   # estimate_mean draws the columns on the device (mlmc_subsample_gather)
        return picked

    # ---- indexing -------------------------------------------------------------------------------------
    def __getitem__(self, key):
        new_qtype, start = self.qtype.get_key(key)
        if not isinstance(self.qtype, qt.ArrayType):
            key = slice(start, start + new_qtype.size())
        parent = self.qtype
        item = Quantity(quantity_type=new_qtype, input_quantities=[self],
                        operation=lambda y: parent._make_getitem_op(y, key=key))
        item._sym = ("getitem", key)
        return item

    def __getattr__(self, name):
        if name.startswith("__") or name in ("qtype", "_memo"):
            raise AttributeError(name)
        static_fun = getattr(self.qtype, name)   # forwards static helpers of the type, e.g. time_interpolation

        def apply_on_quantity(*attr, **d_attr):
            return static_fun(self, *attr, **d_attr)
        return apply_on_quantity

    @staticmethod
    def _concatenate(quantities, qtype, axis=0):
        joined = Quantity(qtype, input_quantities=[*quantities], operation=lambda *chunks: np.concatenate(tuple(chunks), axis=axis))
        joined._sym = ("concat", axis)
        return joined

    @staticmethod
    def _get_base_qtype(args_quantities):
        for q in args_quantities:
            if isinstance(q, Quantity) and type(q.qtype.base_qtype()) == qt.ScalarType:
                return qt.ScalarType()
        return qt.BoolType()

    @staticmethod
    def _method(ufunc, method, *args, **kwargs):
        def _ufunc_call(*chunks):
            return getattr(ufunc, method)(*chunks, **kwargs)
        quantities = [Quantity.wrap(arg) for arg in args]
        from . import lowering
        if lowering.ufunc_is_lowerable(ufunc, method, kwargs):
            # element-wise: the rows broadcast, no need to evaluate the first stored chunk to learn the result size
            rows = max(q.size() for q in quantities)
            qtype = qt.ArrayType(shape=rows, qtype=Quantity._get_base_qtype(quantities))
        else:
            qtype = Quantity._result_qtype(_ufunc_call, quantities)
        result = Quantity(quantity_type=qtype, input_quantities=quantities, operation=_ufunc_call)
        result._sym = ("ufunc", ufunc, method, dict(kwargs))
        return result

    @staticmethod
    def wrap(value):
        if isinstance(value, Quantity):
            return value
        if isinstance(value, bool):
            return QuantityConst(quantity_type=qt.BoolType(), value=value)
        if isinstance(value, (int, float, np.integer, np.floating)):
            return QuantityConst(quantity_type=qt.ScalarType(), value=value)
        if isinstance(value, (list, np.ndarray)):
            value = np.array(value)
            return QuantityConst(quantity_type=qt.ArrayType(shape=value.shape, qtype=qt.ScalarType()), value=value)
        raise ValueError("Values {} are not flat, bool or array (list)".format(value))

    @staticmethod
    def _result_qtype(method, quantities):
        """Result type of a ufunc node, probed on the first stored chunk."""
        chunks = []
        for q in quantities:
            storage = q.get_quantity_storage()
            spec = ChunkSpec() if storage is None else next(storage.chunks())
            chunks.append(q.samples(spec))
        result = method(*chunks)
        return qt.ArrayType(shape=result.shape[0], qtype=Quantity._get_base_qtype(quantities))

    @staticmethod
    def QArray(quantities):
        arr = np.array(quantities)
        flat = arr.flatten()
        return Quantity._concatenate(flat, qtype=qt.ArrayType(arr.shape, Quantity._check_same_qtype(flat)))

    @staticmethod
    def QDict(key_quantity):
        return Quantity._concatenate([q for _, q in key_quantity], qtype=qt.DictType([(k, q.qtype) for k, q in key_quantity]))

    @staticmethod
    def QTimeSeries(time_quantity):
        quantities = [q for _, q in time_quantity]
        times = [t for t, _ in time_quantity]
        return Quantity._concatenate(quantities, qtype=qt.TimeSeriesType(times=times, qtype=Quantity._check_same_qtype(quantities)))

    @staticmethod
    def QField(key_quantity):
        quantities = [q for _, q in key_quantity]
        Quantity._check_same_qtype(quantities)
        return Quantity._concatenate(quantities, qtype=qt.FieldType([(k, q.qtype) for k, q in key_quantity]))

    @staticmethod
    def _check_same_qtype(quantities):
        qtype = quantities[0].qtype
        for q in quantities[1:]:
            if not _same_qtype(qtype, q.qtype):
                raise ValueError("Quantities don't have same QType")
        return qtype


def _same_qtype(a, b):
    """Structural equality of two QTypes (the reference compares by identity, quantity.py:506-512)."""
    if a is b:
        return True
    if type(a) is not type(b) or a.size() != b.size():
        return False
    return True


class QuantityConst(Quantity):
    """Constant operand; stores its value as [M, 1, 1] for broadcasting against chunks (reference: quantity.py:515-565)."""

    def __init__(self, quantity_type, value):
        self.qtype = quantity_type
        self._value = self._process_value(value)
        self._input_quantities = []
        self._selection_id = None
        self._storage = None
        self._memo = (None, {})
        self._sym = ("const",)

    def _process_value(self, value):
        if isinstance(value, (int, float, bool, np.integer, np.floating)):
            value = np.array([value])
        return value[:, np.newaxis, np.newaxis]

    def get_quantity_storage(self):
        return None

    def selection_id(self):
        return self._selection_id

    def _adjust_value(self, value, level_id=None):
        return value

    def samples(self, chunk_spec):
        return self._adjust_value(self._value, chunk_spec.level_id)


class QuantityMean:
    """Result of estimate_mean: per-level means / variances and their MLMC totals (reference: quantity.py:568-651)."""

    def __init__(self, quantity_type, l_means, l_vars, n_samples, n_rm_samples):
        self.qtype = quantity_type
        self._mean = None
        self._var = None
        self._l_means = np.array(l_means)
        self._l_vars = np.array(l_vars)
        self._n_samples = np.array(n_samples)
        self._n_rm_samples = np.array(n_rm_samples)

    def _calculate_mean_var(self):
        self._mean = np.sum(self._l_means, axis=0)
        with np.errstate(all="ignore"):
            self._var = np.sum(self._l_vars / self._n_samples[:, None], axis=0)

    @property
    def mean(self):
        if self._mean is None:
            self._calculate_mean_var()
        return self._reshape(self._mean)

    @property
    def var(self):
        if self._var is None:
            self._calculate_mean_var()
        return self._reshape(self._var)

    @property
    def l_means(self):
        return np.array([self._reshape(m) for m in self._l_means])

    @property
    def l_vars(self):
        return np.array([self._reshape(v) for v in self._l_vars])

    @property
    def n_samples(self):
        return self._n_samples

    @property
    def n_rm_samples(self):
        return self._n_rm_samples

    def _reshape(self, data):
        return self.qtype.reshape(data)

    def __getitem__(self, key):
        new_qtype, start = self.qtype.get_key(key)
        if not isinstance(self.qtype, qt.ArrayType):
            key = slice(start, start + new_qtype.size())
        l_means = self.l_means[:, key]
        l_vars = self.l_vars[:, key]
        return QuantityMean(quantity_type=new_qtype, l_means=l_means.reshape((l_means.shape[0], -1)),
                            l_vars=l_vars.reshape((l_vars.shape[0], -1)), n_samples=self._n_samples,
                            n_rm_samples=self._n_rm_samples)


class QuantityStorage(Quantity):
    """The only node that touches the SampleStorage (reference: quantity.py:654-695)."""

    def __init__(self, storage, qtype):
        self._storage = storage
        self.qtype = qtype
        self._input_quantities = []
        self._operation = None
        self._selection_id = None
        self._memo = (None, {})
        self._sym = ("leaf",)

    def level_ids(self):
        return self._storage.get_level_ids()

    def selection_id(self):
        return id(self)

    def get_quantity_storage(self):
        return self

    def chunks(self, level_id=None):
        return self._storage.chunks(level_id)

    def samples(self, chunk_spec):
        return self._storage.sample_pairs_level(chunk_spec)

    def n_collected(self):
        return self._storage.get_n_collected()
