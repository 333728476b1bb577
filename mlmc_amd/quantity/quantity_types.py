"""Structural types of quantities (reference interface: mlmc/quantity/quantity_types.py:9-246).

A QType describes how the flat leading axis M of a sample chunk [M, n, 2] is organised (dict of time series of
fields of arrays of scalars ...).  Host-side bookkeeping only: no arithmetic on samples happens here apart from
index selection and the time interpolation helper.
"""
import copy

import numpy as np


class QType:
    def __init__(self, qtype):
        self._qtype = qtype

    def size(self) -> int:
        raise NotImplementedError

    def base_qtype(self):
        return self._qtype.base_qtype()

    def replace_scalar(self, substitute_qtype):
        """Copy of this type with the innermost ScalarType replaced (moments / covariance 'at the bottom')."""
        clone = copy.deepcopy(self)
        clone._qtype = self._qtype.replace_scalar(substitute_qtype)
        return clone

    @staticmethod
    def keep_dims(chunk):
        """Chunks always travel as [M, n, 2]: scalars gain a leading axis, deeper arrays are flattened."""
        if chunk.ndim == 2:
            return chunk[np.newaxis]
        if chunk.ndim > 2:
            return chunk.reshape((int(np.prod(chunk.shape[:-2])), chunk.shape[-2], chunk.shape[-1]))
        raise ValueError("Chunk shape not supported")

    def _make_getitem_op(self, chunk, key):
        return QType.keep_dims(chunk[key])

    def reshape(self, data):
        return data


class ScalarType(QType):
    def __init__(self, qtype=float):
        self._qtype = qtype

    def base_qtype(self):
        if isinstance(self._qtype, BoolType):
            return self._qtype.base_qtype()
        return self

    def size(self) -> int:
        return self._qtype.size() if hasattr(self._qtype, 'size') else 1

    def replace_scalar(self, substitute_qtype):
        return substitute_qtype


class BoolType(ScalarType):
    pass


class ArrayType(QType):
    def __init__(self, shape, qtype: QType):
        self._shape = (shape,) if isinstance(shape, (int, np.integer)) else tuple(shape)
        self._qtype = qtype

    def size(self) -> int:
        return int(np.prod(self._shape)) * self._qtype.size()

    def get_key(self, key):
        picked = np.empty(self._shape)[key].shape
        if len(picked) == 1 and picked[0] == 1:      # a single selected item is a scalar
            picked = ()
        if len(picked) > 0:
            return ArrayType(picked, qtype=self._qtype), 0
        return self._qtype, 0

    def _make_getitem_op(self, chunk, key):
        assert self._shape is not None
        shaped = chunk.reshape((*self._shape, chunk.shape[-2], chunk.shape[-1]))
        return QType.keep_dims(shaped[key])

    def reshape(self, data):
        if isinstance(self._qtype, ScalarType):
            return data.reshape(self._shape)
        return data.reshape((*self._shape, int(np.prod(data.shape)) // int(np.prod(self._shape))))


class TimeSeriesType(QType):
    def __init__(self, times, qtype):
        self._times = times.tolist() if isinstance(times, np.ndarray) else list(times)
        self._qtype = qtype

    def size(self) -> int:
        return len(self._times) * self._qtype.size()

    def get_key(self, key):
        if key not in self._times:
            raise KeyError("Item {} was not found in TimeSeries. Available items: {}".format(key, self._times))
        return self._qtype, self._times.index(key) * self._qtype.size()

    @staticmethod
    def time_interpolation(quantity, value):
        """Linear interpolation between the stored times (reference: quantity_types.py:154-167)."""
        from scipy import interpolate
        from . import quantity as qmod
        times = quantity.qtype._times
        inner = quantity.qtype._qtype

        def interp(y):
            cuts = np.arange(1, len(times)) * inner.size()
            parts = np.split(y, cuts, axis=-3)
            return interpolate.interp1d(times, parts, axis=0)(value)
        result = qmod.Quantity(quantity_type=inner, input_quantities=[quantity], operation=interp)
        result._sym = ("interp", tuple(times), value, inner.size())
        return result


class FieldType(QType):
    def __init__(self, args):
        self._dict = dict(args)
        self._qtype = args[0][1]
        assert all(q_type.size() == self._qtype.size() for _, q_type in args)

    def size(self) -> int:
        return len(self._dict) * self._qtype.size()

    def get_key(self, key):
        keys = list(self._dict.keys())
        if key not in keys:
            raise KeyError("Key {} was not found in FieldType. Available keys: {}...".format(key, keys[:5]))
        return self._qtype, keys.index(key) * self._qtype.size()


class DictType(QType):
    def __init__(self, args):
        self._dict = dict(args)
        first = next(iter(self._dict.values())).base_qtype()
        for qtype in list(self._dict.values())[1:]:
            if not isinstance(qtype.base_qtype(), type(first)):
                raise TypeError("qtype {} has base QType {}, expecting {}. All QTypes must have same base QType, "
                                "either SacalarType or BoolType".format(qtype, qtype.base_qtype(), first))

    def base_qtype(self):
        return next(iter(self._dict.values())).base_qtype()

    def size(self) -> int:
        return int(sum(q.size() for q in self._dict.values()))

    def get_qtypes(self):
        return self._dict.values()

    def replace_scalar(self, substitute_qtype):
        return DictType([(k, q.replace_scalar(substitute_qtype)) for k, q in self._dict.items()])

    def get_key(self, key):
        if key not in self._dict:
            raise KeyError("Key {} was not found in DictType. Available keys: {}...".format(key, list(self._dict)[:5]))
        start = 0
        for k, q in self._dict.items():
            if k == key:
                break
            start += q.size()
        return self._dict[key], start
