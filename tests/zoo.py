"""Quantity trees used by the lowering / device-tree tests and by oracle/gen_golden.py (G8).

The same function builds the trees from the reference's objects (when gen_golden.py runs in the build container) and
from mlmc_amd's: it only uses the public Quantity API both sides share (mlmc/quantity/quantity.py)."""
import numpy as np


def result_format(QuantitySpec):
    return [QuantitySpec(name="length", unit="m", shape=(2, 1), times=[1, 2, 3], locations=['10', '20']),
            QuantitySpec(name="width", unit="mm", shape=(2, 1), times=[1, 2, 3], locations=['30', '40'])]


def level_data(n=(700, 500, 300), seed=5):
    """Seeded samples: list over levels of (fine [N, 24], coarse [N, 24] | None)."""
    rng = np.random.default_rng(seed)
    out = []
    for l, nl in enumerate(n):
        fine = rng.normal(size=(nl, 24)) + 2.0
        coarse = fine + 0.1 * rng.normal(size=(nl, 24)) if l else None
        out.append((fine, coarse))
    return out


def expression_zoo(root, Quantity):
    """name -> quantity; the kinds of trees the reference's test/test_quantity_concept.py builds."""
    length = root['length']
    width = root['width']
    loc = length[2]['10']                       # 2 rows
    x = loc[0]
    y = width[1]['30'][1]
    zoo = {
        "leaf_scalar": x,
        "leaf_rows": length[3],                 # 4 rows (two locations)
        "whole_root": root,                     # all 24 rows
        "add_const": x + 1.5,
        "radd_rsub": 3.0 - (2.0 + x),
        "mul_div": (x * y) / (y + 10.0),
        "mod": (x * 7.0) % 3.0,
        "rmod_neg": (-5.0) % (x + 4.0),
        "array_const": loc * np.array([2.0, -1.0]),
        "rows_plus_scalar": length[1] + x,      # 4 rows broadcast with 1
        "central": (x - 2.0) * (x - 2.0),
        "ufunc_sin_exp": np.sin(x) + np.exp(np.negative(y)),
        "ufunc_binary": np.maximum(x, y) - np.minimum(x, 2.0),
        "ufunc_pow_sqrt": np.sqrt(np.abs(x)) + np.power(np.abs(y), 1.5),
        "ufunc_add": np.add(x, y),
        "interp": length.time_interpolation(2.5)['20'],
        "interp_edge": width.time_interpolation(1.0),
        "select_gt": x.select(x > 2.0),
        "select_two": loc.select(loc < 4.5, y >= 1.0),
        "select_expr": (x * y).select(np.logical_or(x > 2.5, y < 1.5)),
        "select_not": x.select(np.logical_not(x > 2.0)),
        "select_vec_mask": loc.select(loc > 0.5),                      # all rows must pass
        "eq_ne": x.select(x != y, y == y),
        "deep": np.log1p(np.abs(np.tanh(x) * np.cos(y) + np.sqrt(np.square(y) + 1.0))) / (1.0 + np.exp2(np.negative(x))),
        "shared_subexpr": (x + y) * (x + y) + (x + y),
    }
    # the reference's QArray wants the very same QType object in every entry (quantity.py:506-512)
    zoo["qarray"] = Quantity.QArray([x, x + 1.0, x * x])
    return zoo
