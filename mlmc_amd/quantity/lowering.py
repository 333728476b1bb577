"""Lowering of a Quantity tree to a register program for the device (mlmc_expr_* in include/mlmc_hip.h).

The reference evaluates a quantity chunk by chunk through a tree of NumPy closures, one temporary [M, n, 2] array per
node (mlmc/quantity/quantity.py:117-135; arithmetic :166-246, ufuncs :366-400, comparisons :250-306, select :137-164,
indexing :327-364, time interpolation quantity_types.py:154-167).  Every one of those nodes acts on each sample
independently, so the whole tree is a function (stored rows of one sample) -> (result rows of that sample, keep flag).
`plan_for(quantity)` writes that function down once as a straight-line program over the stored rows:

    * indexing / dict, field, time keys and concatenations only rename rows (no instruction at all),
    * arithmetic, ufuncs and linear time interpolation become one instruction per result row,
    * a comparison becomes one flag per sample (all rows, fine AND coarse: Quantity._process_mask),
    * `select` contributes its flag to the sample's keep flag; selected samples are compacted on the device.

The kernel (mlmc_amd/csrc/expr.hip) then reads every referenced stored row once and writes only the result rows.
Trees with nodes that are not per-sample functions (sub-sampling, ufunc reductions, user closures) are not lowered
(`plan_for` returns None) and keep the host evaluation of quantity.py.
"""
import ctypes as C
import operator
import struct

import numpy as np

from .. import _lib

# opcode numbers: the enum of include/mlmc_hip.h, in order
_OP_NAMES = ["LOAD", "CONST", "STORE", "SELECT", "ADD", "SUB", "MUL", "DIV", "MOD", "POW", "MAXIMUM", "MINIMUM", "FMAX",
             "FMIN", "ATAN2", "HYPOT", "FMOD", "NEG", "ABS", "SQRT", "SQUARE", "RECIP", "EXP", "EXP2", "EXPM1", "LOG",
             "LOG2", "LOG10", "LOG1P", "SIN", "COS", "TAN", "ASIN", "ACOS", "ATAN", "SINH", "COSH", "TANH", "FLOOR", "CEIL",
             "TRUNC", "RINT", "SIGN", "CBRT", "LT", "LE", "GT", "GE", "EQ", "NE", "AND", "OR", "NOT", "XOR"]
OP = {name: i for i, name in enumerate(_OP_NAMES)}
MAX_REGS = 16
MAX_INSTR = 4096
IMM_A, IMM_B, OP_MASK = 0x4000, 0x8000, 0x07ff      # operand-is-immediate flags of include/mlmc_hip.h
A_PREV, B_PREV, NO_WB = 0x2000, 0x1000, 0x0800      # chaining flags: operand = latest result; result not written back
_IMM_OK = {"ADD", "SUB", "MUL", "DIV", "MOD", "POW", "MAXIMUM", "MINIMUM", "FMAX", "FMIN", "ATAN2", "HYPOT", "FMOD",
           "LT", "LE", "GT", "GE", "EQ", "NE"}

_BINOPS = {operator.add: "ADD", operator.sub: "SUB", operator.mul: "MUL", operator.truediv: "DIV", operator.mod: "MOD"}
_CMPOPS = {operator.lt: "LT", operator.le: "LE", operator.gt: "GT", operator.ge: "GE", operator.eq: "EQ", operator.ne: "NE"}
_UFUNCS2 = {np.add: "ADD", np.subtract: "SUB", np.multiply: "MUL", np.divide: "DIV", np.true_divide: "DIV",
            np.remainder: "MOD", np.mod: "MOD", np.power: "POW", np.float_power: "POW", np.maximum: "MAXIMUM",
            np.minimum: "MINIMUM", np.fmax: "FMAX", np.fmin: "FMIN", np.arctan2: "ATAN2", np.hypot: "HYPOT", np.fmod: "FMOD",
            np.logical_and: "AND", np.logical_or: "OR", np.logical_xor: "XOR"}
_UFUNCS1 = {np.negative: "NEG", np.absolute: "ABS", np.fabs: "ABS", np.sqrt: "SQRT", np.square: "SQUARE",
            np.reciprocal: "RECIP", np.exp: "EXP", np.exp2: "EXP2", np.expm1: "EXPM1", np.log: "LOG", np.log2: "LOG2",
            np.log10: "LOG10", np.log1p: "LOG1P", np.sin: "SIN", np.cos: "COS", np.tan: "TAN", np.arcsin: "ASIN",
            np.arccos: "ACOS", np.arctan: "ATAN", np.sinh: "SINH", np.cosh: "COSH", np.tanh: "TANH", np.floor: "FLOOR",
            np.ceil: "CEIL", np.trunc: "TRUNC", np.rint: "RINT", np.sign: "SIGN", np.cbrt: "CBRT", np.logical_not: "NOT"}
_FLAG_OPS = {"AND", "OR", "XOR", "NOT"}


def ufunc_is_lowerable(ufunc, method, kwargs):
    return method == "__call__" and not kwargs and (ufunc in _UFUNCS1 or ufunc in _UFUNCS2 or ufunc is np.positive)


class NotLowerable(Exception):
    pass


class ExprInstr(C.Structure):
    _fields_ = [("op", C.c_uint16), ("dst", C.c_uint16), ("a", C.c_uint16), ("b", C.c_uint16), ("imm", C.c_double)]


class _Rows(list):
    """Row values of one node; `flag` marks the per-sample result of a comparison."""
    flag = False


class _Builder:
    """SSA construction with common-subexpression elimination; values are indices into self.nodes."""

    def __init__(self):
        self.nodes = []          # (op name, a, b, imm)
        self._cse = {}
        self.in_rows = []        # stored row index of every input slot
        self._slot = {}
        self.select_flags = []

    def _value(self, op, a=-1, b=-1, imm=0.0):
        key = (op, a, b, struct.pack("<d", imm))
        v = self._cse.get(key)
        if v is None:
            v = len(self.nodes)
            self.nodes.append((op, a, b, float(imm)))
            self._cse[key] = v
        return v

    @staticmethod
    def stored(stored_row):
        """A stored row that nobody has computed with yet: it becomes a LOAD (and an input slot) only when used, so
        indexing a 24-row root down to one row reads one row."""
        return ("stored", int(stored_row))

    def mat(self, v):
        if isinstance(v, tuple):
            stored_row = v[1]
            if stored_row not in self._slot:
                self._slot[stored_row] = len(self.in_rows)
                self.in_rows.append(stored_row)
            return self._value("LOAD", self._slot[stored_row])
        return v

    def const(self, value):
        return self._value("CONST", imm=float(value))

    def unary(self, op, a):
        return self._value(op, self.mat(a))

    def binary(self, op, a, b):
        return self._value(op, self.mat(a), self.mat(b))


def _broadcast(a, b):
    if len(a) == len(b):
        return a, b
    if len(a) == 1:
        return a * len(b), b
    if len(b) == 1:
        return a, b * len(a)
    raise NotLowerable("operands with {} and {} rows do not broadcast".format(len(a), len(b)))


def _lower(q, bld, memo):
    got = memo.get(id(q))
    if got is not None:
        return got
    sym = getattr(q, "_sym", None)
    if sym is None or getattr(q, "_volatile", False):
        raise NotLowerable("node {} is not a per-sample function".format(type(q).__name__))
    kind = sym[0]
    ins = [_lower(x, bld, memo) for x in q._input_quantities]
    out = _Rows()
    if kind == "leaf":
        out.extend(bld.stored(r) for r in range(q.qtype.size()))
    elif kind == "const":
        value = np.asarray(q._value)
        if value.ndim != 3 or value.shape[1:] != (1, 1):
            raise NotLowerable("constant of shape {}".format(value.shape))
        out.extend(bld.const(v) for v in value[:, 0, 0].astype(np.float64))
        out.flag = value.dtype == np.bool_
    elif kind == "getitem":
        parent_rows = ins[0]
        probe = np.arange(len(parent_rows), dtype=np.float64).reshape(len(parent_rows), 1, 1)
        picked = q._input_quantities[0].qtype._make_getitem_op(probe, key=sym[1])
        out.extend(parent_rows[int(i)] for i in picked[:, 0, 0])
    elif kind == "concat":
        if sym[1] != 0:
            raise NotLowerable("concatenation along axis {}".format(sym[1]))
        for rows in ins:
            out.extend(rows)
    elif kind == "binop":
        op = _BINOPS.get(sym[1])
        if op is None or ins[0].flag or ins[1].flag:
            raise NotLowerable("binary operation {}".format(sym[1]))
        a, b = _broadcast(ins[0], ins[1])
        out.extend(bld.binary(op, x, y) for x, y in zip(a, b))
    elif kind == "ufunc":
        ufunc, method, kwargs = sym[1], sym[2], sym[3]
        if not ufunc_is_lowerable(ufunc, method, kwargs):
            raise NotLowerable("ufunc {}.{}".format(getattr(ufunc, "__name__", ufunc), method))
        if ufunc is np.positive:
            out.extend(ins[0])
        elif ufunc in _UFUNCS1:
            if len(ins) != 1:
                raise NotLowerable("ufunc arity")
            op = _UFUNCS1[ufunc]
            if (op in _FLAG_OPS) != bool(ins[0].flag):
                raise NotLowerable("logical / arithmetic ufunc applied to the other kind of value")
            out.extend(bld.unary(op, x) for x in ins[0])
            out.flag = ins[0].flag
        else:
            if len(ins) != 2:
                raise NotLowerable("ufunc arity")
            op = _UFUNCS2[ufunc]
            if (op in _FLAG_OPS) != bool(ins[0].flag) or bool(ins[0].flag) != bool(ins[1].flag):
                raise NotLowerable("logical / arithmetic ufunc applied to the other kind of value")
            a, b = _broadcast(ins[0], ins[1])
            out.extend(bld.binary(op, x, y) for x, y in zip(a, b))
            out.flag = ins[0].flag
    elif kind == "cmp":
        op = _CMPOPS.get(sym[1])
        if op is None or ins[0].flag or ins[1].flag:
            raise NotLowerable("comparison {}".format(sym[1]))
        a, b = _broadcast(ins[0], ins[1])
        flag = None
        for x, y in zip(a, b):                       # all rows must satisfy the condition (_process_mask)
            f = bld.binary(op, x, y)
            flag = f if flag is None else bld.binary("AND", flag, f)
        out.append(flag)
        out.flag = True
    elif kind == "select":
        if not ins[1].flag or len(ins[1]) != 1 or ins[0].flag:
            raise NotLowerable("select expects a mask quantity")
        bld.select_flags.append(bld.mat(ins[1][0]))
        out.extend(ins[0])
    elif kind == "interp":
        times, value, inner = list(sym[1]), float(sym[2]), int(sym[3])
        order = np.argsort(times, kind="mergesort")       # scipy.interpolate.interp1d sorts its abscissae
        xs = [float(times[i]) for i in order]
        if not (xs[0] <= value <= xs[-1]):
            raise NotLowerable("time {} outside the stored times (the host evaluation raises)".format(value))
        hi = int(np.clip(np.searchsorted(xs, value), 1, len(xs) - 1))
        lo = hi - 1
        rows = ins[0]
        dx, dt = bld.const(xs[hi] - xs[lo]), bld.const(value - xs[lo])
        for m in range(inner):                           # slope = (y_hi - y_lo) / dx ; y = slope * (t - x_lo) + y_lo
            y_lo = rows[int(order[lo]) * inner + m]
            y_hi = rows[int(order[hi]) * inner + m]
            slope = bld.binary("DIV", bld.binary("SUB", y_hi, y_lo), dx)
            out.append(bld.binary("ADD", bld.binary("MUL", slope, dt), y_lo))
    else:
        raise NotLowerable("unknown node kind {}".format(kind))
    memo[id(q)] = out
    return out


def _schedule(bld, out_rows, chain=True):
    """Order the SSA values row by row (a result row is stored as soon as it is complete, so few values are live at
    once) and assign registers by linear scan.  -> list of (op, dst, a, b, imm), n_regs"""
    order = []            # ("val", v) | ("store", v, row) | ("select", v)
    done = set()

    def is_const(v):
        return v >= 0 and bld.nodes[v][0] == "CONST"

    def imm_form(x):
        """(register operand a, register operand b, flags, imm) of value x when one operand can ride as an immediate."""
        op, a, b, _ = bld.nodes[x]
        if op in _IMM_OK:
            if is_const(b) and not is_const(a):
                return a, -1, IMM_B, bld.nodes[b][3]
            if is_const(a) and not is_const(b):
                return -1, b, IMM_A, bld.nodes[a][3]
        return a, b, 0, 0.0

    def emit(v):
        stack = [(v, False)]
        while stack:
            x, expanded = stack.pop()
            if x in done:
                continue
            op = bld.nodes[x][0]
            deps = [d for d in (imm_form(x)[:2] if op not in ("LOAD", "CONST") else ()) if d >= 0]
            if expanded or not deps:
                done.add(x)
                order.append(("val", x))
            else:
                stack.append((x, True))
                for d in reversed(deps):
                    if d not in done:
                        stack.append((d, False))

    for f in dict.fromkeys(bld.select_flags):
        emit(f)
        order.append(("select", f))
    for row, v in enumerate(out_rows):
        emit(v)
        order.append(("store", v, row))

    def operands(v):
        op = bld.nodes[v][0]
        return () if op in ("LOAD", "CONST") else tuple(d for d in imm_form(v)[:2] if d >= 0)

    # Chaining: the kernel keeps the result of the latest value-producing instruction in VGPRs.  A use of value d at
    # position q is "chained" when d is that latest result (no other producer between); a value all of whose uses are
    # chained needs no LDS register at all.
    latest, producer_before = None, []
    for item in order:
        producer_before.append(latest)
        if item[0] == "val":
            latest = item[1]
    uses = {}                                             # value -> [(position, chained)]
    for pos, item in enumerate(order):
        for d in (operands(item[1]) if item[0] == "val" else (item[1],)):
            uses.setdefault(d, []).append((pos, chain and producer_before[pos] == d))
    no_wb = {v for v, us in uses.items() if all(c for _, c in us)}
    last_use = {v: max(p for p, c in us if not c) for v, us in uses.items() if v not in no_wb}

    free = list(range(MAX_REGS - 1, -1, -1))
    reg = {}
    n_regs = 0
    prog = []
    for pos, item in enumerate(order):
        chained = producer_before[pos] if chain else None  # the value an operand may take from VGPRs here
        if item[0] == "val":
            v = item[1]
            op, a, b, imm = bld.nodes[v]
            flags = 0
            if op == "LOAD":
                ra, rb = a, 0
            elif op == "CONST":
                ra, rb = 0, 0
            else:
                a, b, flags, imm_value = imm_form(v)
                if flags:
                    imm = imm_value
                if a >= 0 and a == chained:
                    flags |= A_PREV
                if b >= 0 and b == chained:
                    flags |= B_PREV
                ra = reg[a] if (a >= 0 and not flags & A_PREV) else 0
                rb = reg[b] if (b >= 0 and not flags & B_PREV) else 0
                for d in {a, b}:                        # operands that die here free their register for the result
                    if d >= 0 and d in reg and last_use.get(d) == pos:
                        free.append(reg.pop(d))
            if v in no_wb or v not in uses:              # lives in VGPRs only (or is never read)
                prog.append((OP[op] | flags | NO_WB, 0, ra, rb, imm))
                continue
            if not free:
                raise NotLowerable("more than {} live values".format(MAX_REGS))
            reg[v] = free.pop()
            n_regs = max(n_regs, reg[v] + 1)
            prog.append((OP[op] | flags, reg[v], ra, rb, imm))
        else:
            v = item[1]
            flags = A_PREV if v == chained else 0
            ra = 0 if flags else reg[v]
            if item[0] == "store":
                prog.append((OP["STORE"] | flags, 0, ra, item[2], 0.0))
            else:
                prog.append((OP["SELECT"] | flags, 0, ra, 0, 0.0))
            if v in reg and last_use.get(v) == pos:
                free.append(reg.pop(v))
    if len(prog) > MAX_INSTR:
        raise NotLowerable("program of {} instructions".format(len(prog)))
    return prog, max(n_regs, 1)


class DevicePlan:
    """A lowered quantity: the stored rows it reads, its program and the device handle of the program."""

    def __init__(self, leaf, in_rows, n_out, prog, n_regs, selects):
        self.leaf = leaf                  # the QuantityStorage node
        self.in_rows = list(in_rows)      # stored row index per input slot
        self.n_out = int(n_out)
        self.prog = prog
        self.n_regs = int(n_regs)
        self.selects = bool(selects)
        self._handle = None
        # two trees with the same program over the same stored rows compute the same rows: cache identity
        self.signature = (tuple(self.prog), tuple(self.in_rows))
        # the tree only picks one stored row (the usual scalar quantity root[name][time][location][i])
        self.is_row_copy = (len(self.prog) == 2 and self.prog[0][0] & OP_MASK == OP["LOAD"]
                            and self.prog[1][0] & OP_MASK == OP["STORE"] and self.n_out == 1)

    def instr_array(self):
        arr = (ExprInstr * len(self.prog))()
        for k, (op, dst, a, b, imm) in enumerate(self.prog):
            arr[k].op, arr[k].dst, arr[k].a, arr[k].b, arr[k].imm = op, dst, a, b, imm
        return arr

    def handle(self):
        if self._handle is None:
            h = C.c_void_p()
            arr = self.instr_array()
            _lib.check(_lib.lib().mlmc_expr_create(arr, len(self.prog), self.n_regs, len(self.in_rows), self.n_out, C.byref(h)))
            self._handle = h
        return self._handle

    def evaluate(self, rows, has_coarse, n, sync=False, sample_stride=None, side_stride=1):
        """rows: device tensors of the stored rows in slot order ([n, 2] interleaved pairs, or [n] at level 0; or views
        into an uploaded storage block [n, 2, M] starting at the row's first value, with sample_stride = 2 M and
        side_stride = M), complete in memory (the caller has synchronised the stream that produced them).
        -> fine [n_out, n'], coarse [n_out, n'] | None as torch CUDA tensors (n' <= n when the quantity selects).
        The kernel runs on the library's stream: the library's own consumers (accumulators) are ordered behind it;
        pass sync=True before reading the tensors with torch."""
        import torch
        dev = rows[0].device
        fine = torch.empty(self.n_out * n, dtype=torch.float64, device=dev)
        coarse = torch.empty(self.n_out * n, dtype=torch.float64, device=dev) if has_coarse else None
        table = (C.c_void_p * len(rows))(*[r.data_ptr() for r in rows])
        n_sel = C.c_int64(n)
        if sample_stride is None:                      # rows uploaded on their own: [n, 2] pairs or [n, 1]
            sample_stride = 2 if has_coarse else 1
        _lib.check(_lib.lib().mlmc_expr_eval(self.handle(), table, 1 if has_coarse else 0, int(n), int(sample_stride),
                                             int(side_stride), fine.data_ptr(),
                                             None if coarse is None else coarse.data_ptr(), C.byref(n_sel)))
        if sync:
            _lib.check(_lib.lib().mlmc_synchronize())
        k = int(n_sel.value)
        fine = fine[:self.n_out * k].view(self.n_out, k)
        if coarse is not None:
            coarse = coarse[:self.n_out * k].view(self.n_out, k)
        return fine, coarse, rows                      # the inputs stay referenced until the caller has finalized

    def kernel_time(self):
        """(ms, launches, algorithmic bytes) of the evaluation kernel since the previous call (needs FLAG_TIMING)."""
        ms, launches, nbytes = C.c_double(), C.c_int64(), C.c_int64()
        _lib.check(_lib.lib().mlmc_expr_kernel_time(self.handle(), C.byref(ms), C.byref(launches), C.byref(nbytes)))
        return ms.value, launches.value, nbytes.value

    JIT_STATES = {0: "interpreter", 2: "compiled", -1: "failed", -2: "not applicable"}

    def jit_state(self):
        """(which form evaluates the program -- "interpreter" | "compiled" | "failed" | "not applicable" --, evaluations of this
        handle that ran the compiled kernel): `mlmc_expr_state`."""
        state, count = C.c_int32(), C.c_int64()
        _lib.check(_lib.lib().mlmc_expr_state(self.handle(), C.byref(state), C.byref(count)))
        return self.JIT_STATES.get(state.value, str(state.value)), count.value

    def __del__(self):
        try:
            if self._handle is not None and _lib._lib is not None:
                _lib._lib.mlmc_expr_destroy(self._handle)
        except Exception:
            pass


def lower(quantity, chain=True):
    """-> DevicePlan of a lowerable tree; raises NotLowerable otherwise.  Needs no GPU.
    chain=False: every value passes through the LDS register file (no MLMC_X_*_PREV / NO_WB flags; tests, tuning)."""
    leaf = quantity.get_quantity_storage()
    if leaf is None:
        raise NotLowerable("quantity without a storage")
    bld = _Builder()
    rows = _lower(quantity, bld, {})
    if rows.flag:
        raise NotLowerable("a mask quantity has no sample rows")
    if len(rows) == 0:
        raise NotLowerable("empty quantity")
    rows = [bld.mat(v) for v in rows]
    if len(bld.in_rows) == 0:
        raise NotLowerable("quantity does not read the storage")
    prog, n_regs = _schedule(bld, rows, chain=chain)
    return DevicePlan(leaf, bld.in_rows, len(rows), prog, n_regs, bool(bld.select_flags))


def plan_for(quantity):
    """Cached DevicePlan of `quantity`, or None when the tree is not lowerable (it is then evaluated on the host)."""
    cached = quantity.__dict__.get("_device_plan", 0)
    if cached != 0:
        return cached
    try:
        plan = lower(quantity)
    except NotLowerable:
        plan = None
    quantity.__dict__["_device_plan"] = plan
    return plan


def run_reference(plan, stored, has_coarse=True):
    """Pure-NumPy interpreter of a program (host logic tests; never used by the product path).
    stored: [M_stored, n, 2|1] -> (values [n_out, n', 2|1], keep [n])"""
    s = stored.shape[-1]
    n = stored.shape[1]
    regs = {}
    out = np.zeros((plan.n_out, n, s))
    keep = np.ones(n, dtype=bool)
    inv = {v: k for k, v in OP.items()}
    table2, table1 = {}, {}
    for k, v in _UFUNCS2.items():
        table2.setdefault(v, k)
    for k, v in _UFUNCS1.items():
        table1.setdefault(v, k)
    prev = None                                           # result of the latest value-producing instruction
    with np.errstate(all="ignore"):
        for op, dst, a, b, imm in plan.prog:
            name = inv[op & OP_MASK]
            va = np.full((n, s), imm) if op & IMM_A else (prev if op & A_PREV else regs.get(a))
            vb = np.full((n, s), imm) if op & IMM_B else (prev if op & B_PREV else regs.get(b))
            if name == "STORE":
                out[b] = va
                continue
            if name == "SELECT":
                keep &= va[:, 0] != 0
                continue
            if name == "LOAD":
                res = stored[plan.in_rows[a]].astype(np.float64)
            elif name == "CONST":
                res = np.full((n, s), imm)
            elif name in ("LT", "LE", "GT", "GE", "EQ", "NE"):
                f = getattr(operator, name.lower())(va, vb).all(axis=1)
                res = np.repeat(f[:, None].astype(np.float64), s, axis=1)
            elif name in ("AND", "OR", "XOR"):
                fn = {"AND": np.logical_and, "OR": np.logical_or, "XOR": np.logical_xor}[name]
                res = fn(va != 0, vb != 0).astype(np.float64)
            elif name == "NOT":
                res = (va == 0).astype(np.float64)
            elif name in table2:
                res = table2[name](va, vb)
            else:
                res = table1[name](va)
            prev = res
            if not op & NO_WB:                            # a result that skips the write-back exists in `prev` only
                regs[dst] = res
    return out[:, keep, :], keep
