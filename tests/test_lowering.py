"""Device lowering of Quantity trees (mlmc_amd/quantity/lowering.py): the program of every lowerable tree, run by a
NumPy interpreter, must reproduce the host evaluation of the tree (which mirrors mlmc/quantity/quantity.py and is
covered against the reference's behaviour in test_host_logic.py).  No GPU needed.  The same expression zoo is run on
the device in tests/test_gpu_api.py::test_device_tree_*."""
import numpy as np
import pytest


def _spec():
    from mlmc_amd.quantity.quantity_spec import QuantitySpec
    from tests.zoo import result_format
    return result_format(QuantitySpec)


def make_storage(n=(700, 500, 300), chunk_size=None, seed=5):
    from mlmc_amd.sample_storage import Memory
    from tests.zoo import level_data
    st = Memory(chunk_size=chunk_size)
    st.save_global_data(result_format=_spec(), level_parameters=[[0.5], [0.1], [0.02]])
    for l, (fine, coarse) in enumerate(level_data(n, seed)):
        st.set_level_samples(l, fine, coarse)
    st.save_n_ops([(l, (10.0 * (l + 1) * nl, nl)) for l, nl in enumerate(n)])
    return st


def expression_zoo(root):
    from mlmc_amd.quantity.quantity import Quantity
    from tests import zoo
    return zoo.expression_zoo(root, Quantity)


def host_chunk(q, chunk):
    from mlmc_amd.quantity import quantity as qmod
    qmod.cache_clear()
    return np.asarray(q.samples(chunk), dtype=np.float64)


def test_programs_reproduce_the_host_tree():
    from mlmc_amd.quantity import lowering
    from mlmc_amd.quantity.quantity import make_root_quantity
    st = make_storage()
    root = make_root_quantity(st, _spec())
    zoo = expression_zoo(root)
    chunks = list(st.chunks())
    for name, q in zoo.items():
        plan = lowering.lower(q)
        assert plan.n_regs <= lowering.MAX_REGS and plan.n_out == q.size(), name
        for chunk in chunks:
            stored = st.sample_pairs_level(chunk)
            want = host_chunk(q, chunk)
            got, keep = lowering.run_reference(plan, stored)
            assert got.shape == want.shape, (name, got.shape, want.shape)
            # same IEEE operations in the same order; libm-backed ufuncs are the same NumPy calls here
            assert np.array_equal(got, want, equal_nan=True), (name, np.nanmax(np.abs(got - want)))
        assert plan.selects == name.startswith(("select", "eq_ne")), name


def test_programs_read_only_the_rows_they_need_and_share_subexpressions():
    from mlmc_amd.quantity import lowering
    from mlmc_amd.quantity.quantity import make_root_quantity
    st = make_storage((50, 40, 30))
    root = make_root_quantity(st, _spec())
    x = root['length'][2]['10'][0]
    y = root['width'][1]['30'][1]
    plan = lowering.lower(x)
    assert plan.in_rows == [4] and len(plan.prog) == 2 and plan.n_regs == 1          # LOAD, STORE
    plan = lowering.lower((x + y) * (x + y) + (x + y))
    names = [lowering._OP_NAMES[p[0] & lowering.OP_MASK] for p in plan.prog]
    assert sorted(plan.in_rows) == [4, 12 + 1] and names.count("ADD") == 2 and names.count("LOAD") == 2
    plan = lowering.lower(root * 2.0)                    # 24 independent rows: stored one by one, few registers
    assert plan.n_out == 24 and plan.n_regs <= 3
    # the instruction array matches the C struct of include/mlmc_hip.h
    import ctypes as C
    assert C.sizeof(lowering.ExprInstr) == 16
    arr = plan.instr_array()
    assert arr[0].op & lowering.OP_MASK in (lowering.OP["LOAD"], lowering.OP["CONST"])


def test_trees_that_are_not_per_sample_functions_fall_back():
    from mlmc_amd.quantity import lowering, quantity_estimate as qe
    from mlmc_amd.quantity.quantity import make_root_quantity
    from mlmc_amd import Legendre
    st = make_storage((50, 40, 30))
    root = make_root_quantity(st, _spec())
    x = root['length'][2]['10'][0]
    assert lowering.plan_for(x.subsample([10, 10, 10])) is None                      # random draw
    assert lowering.plan_for(x.subsample([10, 10, 10]) + 1.0) is None
    assert lowering.plan_for(np.add.reduce(root['length'][2]['10'], axis=0)) is None  # ufunc method other than __call__
    assert lowering.plan_for(qe.moment(x, Legendre(3, (0.0, 4.0)), 1)) is None        # user closure
    with pytest.raises(lowering.NotLowerable):
        lowering.lower(root['length'].time_interpolation(7.0))                         # outside the stored times
    with pytest.raises(lowering.NotLowerable):
        lowering.lower(x > 1.0)                                                        # a mask has no sample rows
    # too many live values for the register file: 20 distinct terms all needed by the last row
    terms = [np.sin(x + float(k)) for k in range(20)]
    prod = terms[0]
    for t in terms[1:]:
        prod = prod * t
    assert lowering.plan_for(prod) is not None           # a chain: each term dies at once
    from mlmc_amd.quantity.quantity import Quantity
    rev = Quantity.QArray(terms + [terms[k] + terms[19 - k] for k in range(20)])
    assert lowering.plan_for(rev) is not None or True    # may or may not fit; must not raise


def test_header_opcodes_match_the_python_table():
    import os
    import re
    from mlmc_amd.quantity import lowering
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "mlmc_hip.h")).read()
    body = hdr[hdr.index("MLMC_X_LOAD = 0"):hdr.index("MLMC_X_N_OPS")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = re.findall(r"MLMC_X_([A-Z0-9]+)", body)
    assert names == lowering._OP_NAMES
    assert int(re.search(r"MLMC_EXPR_MAX_REGS (\d+)", hdr).group(1)) == lowering.MAX_REGS
    assert int(re.search(r"MLMC_EXPR_MAX_INSTR (\d+)", hdr).group(1)) == lowering.MAX_INSTR


def _random_tree(rng, leaves, depth):
    """Random per-sample expression over the given scalar / vector quantities (operators the reference's Quantity has)."""
    if depth == 0 or rng.random() < 0.15:
        return leaves[rng.integers(len(leaves))]
    kind = rng.integers(7)
    a = _random_tree(rng, leaves, depth - 1)
    if kind == 0:
        return a + float(rng.normal())
    if kind == 1:
        return float(rng.normal()) - a
    if kind == 2:
        b = _random_tree(rng, leaves, depth - 1)
        return a * b if a.size() == b.size() or min(a.size(), b.size()) == 1 else a * 0.5
    if kind == 3:
        return a / (np.abs(_scalar(rng, leaves, depth - 1)) + 1.0)
    if kind == 4:
        f = [np.sqrt, np.square, np.negative, np.floor, np.sign][rng.integers(5)]
        return f(np.abs(a)) if f is np.sqrt else f(a)
    if kind == 5:
        b = _scalar(rng, leaves, depth - 1)
        return [np.maximum, np.minimum, np.fmax, np.subtract][rng.integers(4)](a, b)
    return a % (np.abs(_scalar(rng, leaves, depth - 1)) + 0.5)


def _scalar(rng, leaves, depth):
    q = _random_tree(rng, leaves, depth)
    return q if q.size() == 1 else q[0]


@pytest.mark.parametrize("seed", range(12))
def test_random_trees_lower_to_equivalent_programs(seed):
    from mlmc_amd.quantity import lowering
    from mlmc_amd.quantity.quantity import make_root_quantity
    rng = np.random.default_rng(1000 + seed)
    st = make_storage((120, 90, 60), seed=seed)
    root = make_root_quantity(st, _spec())
    leaves = [root['length'][1]['10'][0], root['length'][2]['20'][1], root['width'][3]['30'][0], root['width'][2]['40'],
              root['length'].time_interpolation(1.25)['10']]
    for _ in range(6):
        q = _random_tree(rng, leaves, depth=4)
        if rng.random() < 0.5:                           # half of the trees also select samples
            m1 = _scalar(rng, leaves, 2) > float(rng.normal() + 2.0)
            m2 = _scalar(rng, leaves, 2) <= float(rng.normal() + 3.0)
            q = q.select(m1, m2) if rng.random() < 0.5 else q.select(np.logical_or(m1, m2))
        plan = lowering.plan_for(q)
        if plan is None:                                 # more live values than registers: host path, nothing to compare
            continue
        for chunk in st.chunks():
            with np.errstate(all="ignore"):
                want = host_chunk(q, chunk)
                got, _ = lowering.run_reference(plan, st.sample_pairs_level(chunk))
            assert got.shape == want.shape
            assert np.array_equal(got, want, equal_nan=True), (seed, np.nanmax(np.abs(got - want)))


def _g8():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "G8_quantity_tree.npz"))


def test_reference_quantity_tree_golden_host_and_program():
    """G8_quantity_tree.npz: chunks of the zoo's trees evaluated by the REFERENCE's Quantity classes
    (oracle/gen_golden.py g8).  The host tree of mlmc_amd and the lowered programs (NumPy interpreter) reproduce them bit
    for bit -- the device kernel is checked against the same file in tests/test_gpu_api.py."""
    from mlmc_amd.quantity import lowering
    from mlmc_amd.quantity.quantity import make_root_quantity
    g8 = _g8()
    st = make_storage(tuple(int(v) for v in g8["n"]), seed=int(g8["seed"]))
    root = make_root_quantity(st, _spec())
    n_checked = 0
    for name, q in expression_zoo(root).items():
        plan = lowering.lower(q)
        for chunk in st.chunks():
            want = g8["{}__L{}".format(name, chunk.level_id)]
            with np.errstate(all="ignore"):
                host = host_chunk(q, chunk)
                prog, _ = lowering.run_reference(plan, st.sample_pairs_level(chunk))
            assert host.shape == want.shape and np.array_equal(host, want, equal_nan=True), (name, "host tree")
            assert prog.shape == want.shape and np.array_equal(prog, want, equal_nan=True), (name, "program")
            n_checked += 1
    assert n_checked == len(g8.files) - 2


def test_chained_programs_keep_values_in_vgprs():
    """The scheduler flags operands that are the latest result (A_PREV / B_PREV) and drops the LDS write-back of values
    read only that way (NO_WB): fewer registers, same rows as the unchained program (both run through the NumPy
    interpreter, which keeps a skipped value in `prev` only)."""
    from mlmc_amd.quantity import lowering
    from mlmc_amd.quantity.quantity import make_root_quantity
    st = make_storage((50, 40, 30))
    root = make_root_quantity(st, _spec())
    x = root['length'][2]['10'][0]
    y = root['width'][1]['30'][1]
    q = (x - 0.1) * (x - 0.1) / (np.abs(y) + 1.0)
    plain, chained = lowering.lower(q, chain=False), lowering.lower(q)
    flags = lowering.A_PREV | lowering.B_PREV | lowering.NO_WB
    assert not any(p[0] & flags for p in plain.prog) and plain.n_regs == 2
    assert chained.n_regs == 1 and len(chained.prog) == len(plain.prog)
    written = [p for p in chained.prog if not p[0] & lowering.NO_WB
               and p[0] & lowering.OP_MASK not in (lowering.OP["STORE"], lowering.OP["SELECT"])]
    assert len(written) == 1                                   # only (x - 0.1)^2 outlives its successor
    assert chained.signature != plain.signature
    for name, tree in expression_zoo(root).items():
        a, b = lowering.lower(tree, chain=False), lowering.lower(tree)
        assert b.n_regs <= a.n_regs, name
        for chunk in st.chunks():
            stored = st.sample_pairs_level(chunk)
            (va, ka), (vb, kb) = lowering.run_reference(a, stored), lowering.run_reference(b, stored)
            assert np.array_equal(va, vb, equal_nan=True) and np.array_equal(ka, kb), name
