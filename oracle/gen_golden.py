#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY -- generate tests/golden/*.npz|json by IMPORTING the reference.

Runs only in the build container, where /root/reference exists (it does not exist on the GPU
box; nothing at test/bench time reads it).  The reference's Python modules are imported
unmodified; nothing from them is copied into this repo -- the fixtures hold inputs and the
reference's outputs only.

Harness-side accommodations (no edits to the reference):
  * ``mlmc`` is pre-registered as a namespace stub so that ``mlmc/__init__.py`` (which imports
    the absent h5py) is never executed; sub-modules are imported unmodified.
  * ``memoization.cached`` (absent third-party package, a pure result cache with no arithmetic)
    is replaced by a pass-through decorator that offers ``.cache_clear()``; used only for the
    fixtures that go through ``mlmc.quantity`` / ``mlmc.estimator`` (G2, G3, G4, G7, G8, G9).
    G1, G5, G6 need no stand-in at all.
  * ``np.float = float`` (alias removed in NumPy 1.24, used by sample_storage.py:174).
  * G9 only: ``Memory._save_successful`` is fed an object array (``np.array`` of ragged tuples raises on NumPy >= 1.24,
    sample_storage.py:171) and the cost per sample is ``n_ops_estimate(step)`` instead of the pool's measured wall time.

Usage:  python oracle/gen_golden.py   (writes tests/golden/)
"""
import hashlib
import json
import os
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _install_shims():
    m = types.ModuleType("mlmc")
    m.__path__ = [os.path.join(REF, "mlmc")]
    sys.modules["mlmc"] = m

    memo = types.ModuleType("memoization")

    def cached(custom_key_maker=None, **_kw):
        def deco(fn):
            def wrapper(*a, **k):
                return fn(*a, **k)
            wrapper.cache_clear = lambda: None
            return wrapper
        return deco
    memo.cached = cached
    sys.modules["memoization"] = memo
    if not hasattr(np, "float"):
        np.float = float
    import matplotlib
    matplotlib.use("Agg")


def synth_level(level, n, steps, seed=1234, loc=0.0, scale=1.0):
    """Synthetic level data, formula of mlmc/sim/synth_simulation.py:37-46 (same as oracle_np.synth_level_samples)."""
    rng = np.random.default_rng(seed + level)
    x = loc + scale * rng.standard_normal(n)
    root = np.sqrt(1e-4 + np.abs(x))
    fine = x + steps[level] * root
    coarse = np.zeros(n) if level == 0 else x + steps[level - 1] * root
    return fine, coarse


def make_storage(level_arrays, level_parameters, n_ops, q_specs):
    """Reference Memory storage filled directly: _results[level] = ndarray [N, 2, M]."""
    from mlmc.sample_storage import Memory
    st = Memory()
    st.save_global_data(result_format=q_specs, level_parameters=level_parameters)
    for l, arr in enumerate(level_arrays):
        st._results[l] = np.ascontiguousarray(arr)
        st._n_finished[l] = arr.shape[0]
    st._n_ops = {l: v for l, v in enumerate(n_ops)}
    st.get_n_ops = lambda: [st._n_ops[l] for l in range(len(level_arrays))]
    return st


def g1_basis():
    import mlmc.moments as mm
    out = {}
    eps = np.finfo(float).eps
    cases = []
    dom = (-3.5, 4.25)
    grid = np.concatenate([
        np.linspace(dom[0], dom[1], 41),
        [np.nextafter(dom[0], -np.inf), np.nextafter(dom[1], np.inf), dom[0], dom[1], -1e-300, 0.0, 1e-300,
         np.nan, np.inf, -np.inf, 17.0, -17.0, 0.1234567890123]])
    out["grid"] = grid
    for R in (1, 2, 5, 10, 32, 64):
        for safe in (True, False):
            g = grid if safe else np.where(np.isfinite(grid), grid, 0.5)
            fn = mm.Legendre(R, dom, safe_eval=safe)
            with np.errstate(all="ignore"):
                out[f"legendre_R{R}_safe{int(safe)}"] = fn.eval_all(g)
            fn = mm.Monomial(R, dom, safe_eval=safe)
            with np.errstate(all="ignore"):
                out[f"monomial_R{R}_safe{int(safe)}"] = fn.eval_all(g)
    out["grid_nosafe"] = np.where(np.isfinite(grid), grid, 0.5)
    for R in (1, 2, 5, 6, 33):
        fn = mm.Fourier(R, dom)
        with np.errstate(all="ignore"):
            out[f"fourier_R{R}_safe1"] = fn.eval_all(grid)
    # log transform, incl. non-positive inputs
    ldom = (0.05, 30.0)
    lgrid = np.concatenate([np.geomspace(ldom[0], ldom[1], 33), [0.0, -1.0, 1e-320, 0.049999, 30.0000001, np.nan, np.inf]])
    out["lgrid"] = lgrid
    out["ldom"] = np.array(ldom)
    out["dom"] = np.array(dom)
    for R in (5, 32):
        with np.errstate(all="ignore"):
            out[f"legendre_log_R{R}"] = mm.Legendre(R, ldom, log=True).eval_all(lgrid)
            out[f"monomial_log_R{R}"] = mm.Monomial(R, ldom, log=True).eval_all(lgrid)
    # custom ref_domain
    with np.errstate(all="ignore"):
        out["legendre_ref_R7"] = mm.Legendre(7, dom, ref_domain=(-0.5, 0.75)).eval_all(grid)
        out["monomial_ref_R7"] = mm.Monomial(7, dom, ref_domain=(-1.0, 2.0)).eval_all(grid)
    # 3-D input [M, n, 2]
    rng = np.random.default_rng(7)
    x3 = rng.normal(size=(3, 17, 2)) * 2.0
    out["x3"] = x3
    with np.errstate(all="ignore"):
        out["legendre_x3_R9"] = mm.Legendre(9, dom).eval_all(x3)
    # transformed moments
    mat = rng.normal(size=(6, 9))
    mat[0, :] = 0
    mat[0, 0] = 1
    out["tm_matrix"] = mat
    with np.errstate(all="ignore"):
        out["transformed_x3"] = mm.TransformedMoments(mm.Legendre(9, dom), mat).eval_all(x3)
        out["transformed_grid_size4"] = mm.TransformedMoments(mm.Legendre(9, dom), mat).eval_all(grid, 4)
    # eval_single_moment / eval / derivative bases (API completeness, a6/a8)
    fn = mm.Legendre(6, dom)
    with np.errstate(all="ignore"):
        out["legendre_single3"] = fn.eval_single_moment(3, grid)
        out["legendre_eval3"] = fn.eval(3, grid)
        out["legendre_diff"] = fn.eval_diff(grid)
        out["legendre_diff2"] = fn.eval_diff2(grid)
        out["legendre_der1"] = fn.eval_all_der(grid, degree=1)
        out["monomial_eval3"] = mm.Monomial(6, dom).eval(3, grid)
    # reference's own known-answer cases (test/test_moments.py:61-70, :44-58, :16-30)
    v = np.array([0.0, 0.25, 0.5, 0.75, 1.0])
    out["kat_legendre"] = mm.Legendre(4, (-1.0, 1.0))(v)
    out["kat_fourier"] = mm.Fourier(6, (0, 1))(v)
    v2 = np.array([-2, -1, -0.5, 0, 0.5, 1, 2])
    out["kat_monomial"] = mm.Monomial(5, safe_eval=False)(v2)
    np.savez_compressed(os.path.join(OUT, "G1_basis.npz"), **out)
    print("G1", len(out))


def _scalar_spec():
    from mlmc.quantity.quantity_spec import QuantitySpec
    return [QuantitySpec(name="q", unit="m", shape=(1, 1), times=[1], locations=['0'])]


def _vec_spec():
    from mlmc.quantity.quantity_spec import QuantitySpec
    return [QuantitySpec(name="q", unit="m", shape=(2, 1), times=[1, 2], locations=['0'])]   # M = 4


def _levels(L, N, steps, M=1, seed=1234, nan_every=0, loc=0.0, scale=1.0):
    arrs = []
    for l in range(L):
        fine, coarse = synth_level(l, N[l], steps, seed=seed, loc=loc, scale=scale)
        a = np.empty((N[l], 2, M))
        for m in range(M):
            a[:, 0, m] = fine + 0.125 * m
            a[:, 1, m] = coarse + (0.125 * m if l > 0 else 0.0)
        if nan_every:
            a[::nan_every, 0, 0] = np.nan
            if l > 0:
                a[3::nan_every * 2, 1, M - 1] = np.nan
        arrs.append(a)
    return arrs


def g2_g3_g4():
    import mlmc.moments as mm
    import mlmc.quantity.quantity as q
    import mlmc.quantity.quantity_estimate as qe
    import mlmc.estimator as est
    import scipy.stats

    dom = tuple(scipy.stats.norm().ppf([1e-4, 1 - 1e-4]))
    out2, out3, out4 = {}, {}, {}
    out2["domain"] = np.array(dom)
    for tag, L, N, M, nan_every in (("L3", 3, [4000, 3000, 2000], 1, 0),
                                    ("L5", 5, [20000, 9000, 5000, 3000, 1500], 1, 0),
                                    ("L3nan", 3, [1500, 1200, 900], 1, 7),
                                    ("L3M4", 3, [800, 700, 600], 4, 11),
                                    ("L1", 1, [5000], 1, 0)):
        steps_ll = est.determine_level_parameters(L, [0.5, 0.01]) if L > 1 else [[0.01]]
        steps = [s[0] for s in steps_ll]
        arrs = _levels(L, N, steps, M=M, nan_every=nan_every)
        n_ops = [(1 / h) ** 2 * np.log(max(1 / h, 2.0)) for h in steps]
        spec = _scalar_spec() if M == 1 else _vec_spec()
        st = make_storage(arrs, steps_ll, n_ops, spec)
        root = q.make_root_quantity(st, spec)
        quantity = root['q'][1]['0'] if M == 1 else root['q']      # scalar-array [1,1] / time series
        if M == 1:
            quantity = quantity[0, 0]
        out2[f"{tag}_N"] = np.array(N)
        out2[f"{tag}_steps"] = np.array(steps)
        out2[f"{tag}_M"] = np.array(M)
        out2[f"{tag}_nan_every"] = np.array(nan_every)
        for R in (5, 10, 32, 64):
            if M > 1 and R > 5:
                continue
            fn = mm.Legendre(R, dom)
            for bottom in (True, False):
                with np.errstate(all="ignore"):
                    r = qe.estimate_mean(qe.moments(quantity, fn, mom_at_bottom=bottom))
                key = f"{tag}_leg{R}_b{int(bottom)}"
                out2[key + "_mean"] = r.mean
                out2[key + "_var"] = r.var
                out2[key + "_l_means"] = r.l_means
                out2[key + "_l_vars"] = r.l_vars
                out2[key + "_n"] = r.n_samples
                out2[key + "_n_rm"] = r.n_rm_samples
        if M == 1:
            fn = mm.Monomial(6, dom)
            with np.errstate(all="ignore"):
                r = qe.estimate_mean(qe.moments(quantity, fn))
            out2[f"{tag}_mono6_mean"] = r.mean
            out2[f"{tag}_mono6_var"] = r.var
            out2[f"{tag}_mono6_n"] = r.n_samples
            out2[f"{tag}_mono6_n_rm"] = r.n_rm_samples
            # plain mean of the quantity (no moments node)
            with np.errstate(all="ignore"):
                r = qe.estimate_mean(quantity)
            out2[f"{tag}_plain_mean"] = r.mean
            out2[f"{tag}_plain_var"] = r.var
            out2[f"{tag}_plain_l_means"] = r.l_means
            out2[f"{tag}_plain_l_vars"] = r.l_vars
            out2[f"{tag}_plain_n"] = r.n_samples
            out2[f"{tag}_plain_n_rm"] = r.n_rm_samples
            # single moment (qe.moment)
            with np.errstate(all="ignore"):
                r = qe.estimate_mean(qe.moment(quantity, mm.Legendre(8, dom), 3))
            out2[f"{tag}_moment3_mean"] = r.mean
            out2[f"{tag}_moment3_var"] = r.var
        # covariance (small N only: the reference materialises [2, M, N, R, R])
        if tag in ("L3", "L3nan", "L1"):
            for R in (8, 16, 24, 64):
                Ncov = [min(n, 1500 if R < 64 else 300) for n in N]
                arrs_c = [a[:n] for a, n in zip(arrs, Ncov)]
                stc = make_storage(arrs_c, steps_ll, n_ops, spec)
                rootc = q.make_root_quantity(stc, spec)
                qc = rootc['q'][1]['0'][0, 0]
                e = est.Estimate(qc, stc, mm.Legendre(R, dom))
                with np.errstate(all="ignore"):
                    r = qe.estimate_mean(qe.covariance(qc, mm.Legendre(R, dom)))
                key = f"{tag}_cov{R}"
                out3[key + "_Ncov"] = np.array(Ncov)
                out3[key + "_mean"] = r.mean
                out3[key + "_var"] = r.var
                out3[key + "_l_means"] = r.l_means
                out3[key + "_l_vars"] = r.l_vars
                out3[key + "_n"] = r.n_samples
                out3[key + "_n_rm"] = r.n_rm_samples
        # regression + allocation
        if M == 1 and nan_every == 0:
            for R in (5, 32):
                fn = mm.Legendre(R, dom)
                e = est.Estimate(quantity, st, fn)
                with np.errstate(all="ignore"):
                    raw_vars, n_s = e.estimate_diff_vars(fn)
                    if L >= 3:
                        # the chi2 quad at estimator.py:111 is a dead value; skip its 0.2 s by pre-seeding the cache
                        e._saved_var_var = (list(N), np.ones(L))
                    reg_vars, n_ops_r = e.estimate_diff_vars_regression(N, fn)
                    n_est = est.estimate_n_samples_for_target_variance(1e-6, reg_vars, n_ops_r, n_levels=L)
                    n_est_raw = est.estimate_n_samples_for_target_variance(1e-5, raw_vars, n_ops_r, n_levels=L)
                out4[f"{tag}_R{R}"] = dict(raw_vars=np.asarray(raw_vars).tolist(), steps=steps, n_ops=list(map(float, n_ops)),
                                           reg_vars=np.asarray(reg_vars).tolist(), n_estimated=n_est.tolist(),
                                           n_estimated_raw=n_est_raw.tolist(), n_samples=np.asarray(n_s).tolist())
    out2["seed"] = np.array(1234)
    np.savez_compressed(os.path.join(OUT, "G2_estimate_mean.npz"), **out2)
    np.savez_compressed(os.path.join(OUT, "G3_cov.npz"), **out3)
    # misc free functions
    out4["level_params_5"] = est.determine_level_parameters(5, [0.5, 0.01])
    out4["level_params_1"] = est.determine_level_parameters(1, [0.5, 0.01])
    out4["determine_n_samples_5"] = est.determine_n_samples(5).tolist()
    out4["determine_n_samples_4_1000_10"] = est.determine_n_samples(4, [1000, 10]).tolist()
    with open(os.path.join(OUT, "G4_alloc.json"), "w") as f:
        json.dump(out4, f)
    print("G2", len(out2), "G3", len(out3), "G4", len(out4))


def _exact_cov(fn, pdf):
    import mlmc.tool.simple_distribution as sd
    return sd.compute_semiexact_cov(fn, pdf)


def g5_g6():
    import mlmc.moments as mm
    import mlmc.tool.simple_distribution as sd
    import mlmc.tool.distribution as dd
    import scipy.stats as stats

    out5, out6 = {}, {}
    for name, distr, quant in (("norm12", stats.norm(loc=1, scale=2), 0.01),
                               ("norm110", stats.norm(loc=1, scale=10), 0.01),
                               ("lognorm", stats.lognorm(scale=np.exp(1), s=1), 0.01)):
        domain = tuple(distr.ppf([quant, 1 - quant]))
        # truncated, renormalised density as in test/test_distribution.py (CutDistribution)
        norm_c = distr.cdf(domain[1]) - distr.cdf(domain[0])

        def pdf(x, distr=distr, norm_c=norm_c):
            return distr.pdf(x) / norm_c
        for R in (7, 21, 41):
            base = mm.Legendre(R, domain)
            cov = sd.compute_semiexact_cov(base, pdf)
            key = f"{name}_R{R}"
            out5[key + "_cov"] = cov
            out5[key + "_domain"] = np.array(domain)
            for tol in (1e-4, 0.0, 1e-10):
                ortho, (ev, thr, L_mn) = sd.construct_ortogonal_moments(base, cov, tol)
                tk = key + "_tol{:g}".format(tol)
                out5[tk + "_eval"] = ev
                out5[tk + "_threshold"] = np.array(thr)
                out5[tk + "_L"] = L_mn
            # tol=None: threshold from the change of slope of the log-eigenvalues (simple_distribution.py:781-782, :584-609).
            # An exact covariance has eigenvalues down at rounding level (some negative); covariances estimated from samples
            # -- the case the rule is made for -- are emulated by a small symmetric perturbation with a fixed seed.
            for tag, noise in (("none", 0.0), ("none_n6", 1e-6), ("none_n4", 1e-4)):
                rng = np.random.default_rng(20 + R)
                E = rng.standard_normal(cov.shape)
                cov_n = cov + noise * (E + E.T) / 2
                cov_n[0, :] = cov[0, :]
                cov_n[:, 0] = cov[:, 0]
                tk = key + "_tol" + tag
                try:
                    ortho_n, (ev, thr, L_mn) = sd.construct_ortogonal_moments(base, cov_n.copy(), None)
                except Exception as e:       # recorded: the port has to fail the same way
                    out5[tk + "_error"] = np.array(type(e).__name__)
                    continue
                out5[tk + "_cov"] = cov_n
                out5[tk + "_eval"] = ev
                out5[tk + "_threshold"] = np.array(thr)
                out5[tk + "_L"] = L_mn
            ortho, (ev, thr, L_mn) = sd.construct_ortogonal_moments(base, cov, 1e-4 if name != "lognorm" else 1e-3)
            exact_moments = sd.compute_semiexact_moments(ortho, pdf)
            moment_data = np.stack([exact_moments, np.ones_like(exact_moments)], axis=1)
            out6[key + "_L"] = L_mn
            out6[key + "_domain"] = np.array(domain)
            out6[key + "_moment_data"] = moment_data
            d = sd.SimpleDistribution(ortho, moment_data.copy(), domain=domain)
            res = d.estimate_density_minimize(tol=1e-8)
            xg = np.linspace(domain[0], domain[1], 401)
            out6[key + "_sd_multipliers"] = d.multipliers
            out6[key + "_sd_nit"] = np.array(res.nit)
            out6[key + "_sd_fun_norm"] = np.array(res.fun_norm)
            out6[key + "_sd_success"] = np.array(bool(res.success))
            out6[key + "_xgrid"] = xg
            out6[key + "_sd_density"] = d.density(xg)
            out6[key + "_sd_cdf"] = d.cdf(xg[::8])
            out6[key + "_exact_pdf"] = pdf(xg)
            out6[key + "_sd_nquad"] = np.array(len(d._quad_points))
            # diagnostics of simple_distribution.py:330-464 on the reference's own reconstruction
            kl = sd.KL_divergence(pdf, lambda x: float(d.density(x)[0]), domain[0], domain[1])
            out6[key + "_sd_KL"] = np.array(kl)
            out6[key + "_sd_L2"] = np.array(sd.L2_distance(pdf, lambda x: float(d.density(x)[0]), domain[0], domain[1]))
            if R == 7:
                out6[key + "_exact_moments"] = sd.compute_exact_moments(base, pdf)
                out6[key + "_exact_cov"] = sd.compute_exact_cov(base, pdf)
                out6[key + "_semiexact_moments_base"] = sd.compute_semiexact_moments(base, pdf)
                out6[key + "_old_exact_moments"] = dd.compute_exact_moments(base, pdf)
            print("G6", key, "nit", res.nit, "fun_norm", res.fun_norm, "Q", len(d._quad_points), "KL", kl)
        # older solver (tool/distribution.py) on plain Legendre moments with small noise-free data
        for R in (5, 11):
            base = mm.Legendre(R, domain)
            exact = sd.compute_semiexact_moments(base, pdf)
            md = np.stack([exact, np.full_like(exact, 1e-6)], axis=1)
            d = dd.Distribution(base, md.copy(), domain=domain, force_decay=(True, True))
            res = d.estimate_density_minimize(tol=1e-6, reg_param=0.0)
            key = f"{name}_old_R{R}"
            xg = np.linspace(domain[0], domain[1], 201)
            out6[key + "_moment_data"] = md
            out6[key + "_domain"] = np.array(domain)
            out6[key + "_multipliers"] = d.multipliers
            out6[key + "_moment_errs"] = d._moment_errs
            out6[key + "_nit"] = np.array(res.nit)
            out6[key + "_fun_norm"] = np.array(res.fun_norm)
            out6[key + "_xgrid"] = xg
            out6[key + "_density"] = d.density(xg)
            out6[key + "_end_diff"] = np.dot(d._end_point_diff, res.x)       # > 0: the decay penalty is active at the solution
            out6[key + "_KL"] = np.array(dd.KL_divergence(pdf, lambda x: float(d.density(x)[0]), domain[0], domain[1]))
            out6[key + "_L2"] = np.array(dd.L2_distance(pdf, lambda x: float(d.density(x)[0]), domain[0], domain[1]))
            print("G6 old", key, "nit", res.nit, "fun_norm", res.fun_norm)
    np.savez_compressed(os.path.join(OUT, "G5_ortho.npz"), **out5)
    np.savez_compressed(os.path.join(OUT, "G6_maxent.npz"), **out6)
    print("G5", len(out5), "G6", len(out6))


def g7_chain():
    """Sample-id -> md5 seed -> SynthSimulation.calculate -> Legendre(5) means; reproduces the reference's
    own golden vector ref_means of test/test_sampling_pools.py:18 (3 levels x 10 samples, norm(1,2))."""
    ruamel = types.ModuleType("ruamel")
    ruamel.yaml = types.ModuleType("ruamel.yaml")
    sys.modules.setdefault("ruamel", ruamel)
    sys.modules.setdefault("ruamel.yaml", ruamel.yaml)
    import scipy.stats as stats
    import mlmc.moments as mm
    import mlmc.quantity.quantity as q
    import mlmc.estimator as est
    from mlmc.sim.synth_simulation import SynthSimulation
    from mlmc.sampling_pool import SamplingPool

    step_range = [[0.01], [0.001], [0.0001]]
    distr = stats.norm(loc=1, scale=2)
    sim = SynthSimulation(dict(distr=distr, complexity=2, nan_fraction=0))
    fmt = sim.result_format()
    out = {"levels": []}
    arrs = []
    for l in range(3):
        fine_step = step_range[l][0]
        coarse_step = step_range[l - 1][0] if l > 0 else 0
        cfg = dict(sim.config)
        cfg["fine"] = {"step": fine_step}
        cfg["coarse"] = {"step": coarse_step}
        cfg["res_format"] = fmt
        ids, seeds, fines, coarses = [], [], [], []
        for i in range(10):
            sid = "L{:02d}_S{:07d}".format(l, i)
            seed = SamplingPool.compute_seed(sid)
            assert seed == np.frombuffer(hashlib.md5(sid.encode('ascii')).digest(), dtype='uint32')[0]
            f, c = SynthSimulation.calculate(cfg, seed)
            ids.append(sid)
            seeds.append(int(seed))
            fines.append(f)
            coarses.append(c)
        arr = np.stack([np.array(fines), np.array(coarses)], axis=1)      # [N, 2, M]
        arrs.append(arr)
        out["levels"].append(dict(sample_ids=ids, seeds=seeds, fine=np.array(fines).tolist(),
                                  coarse=np.array(coarses).tolist()))
    n_ops = [sim.n_ops_estimate(s[0]) for s in step_range]
    st = make_storage(arrs, step_range, n_ops, fmt)
    root = q.make_root_quantity(st, fmt)
    value_quantity = root['length'][1]['10'][0]
    true_domain = distr.ppf([0.0001, 0.9999])
    fn = mm.Legendre(5, true_domain)
    e = est.Estimate(value_quantity, st, fn)
    means, vars_ = e.estimate_moments(fn)
    ref_means = [1., -0.03814235, -0.42411443, 0.05103307, 0.2123083]     # test/test_sampling_pools.py:18
    assert np.allclose(ref_means, means, atol=1e-5), means
    assert means[0] == 1 and vars_[0] == 0
    out["domain"] = list(map(float, true_domain))
    out["means"] = np.asarray(means).tolist()
    out["vars"] = np.asarray(vars_).tolist()
    out["ref_means_test_sampling_pools_py_18"] = ref_means
    out["selection"] = "root['length'][1]['10'][0]  -> flat index 0 of M=24"
    with open(os.path.join(OUT, "G7_chain.json"), "w") as f:
        json.dump(out, f)
    print("G7 means", means)


def g8_quantity_tree():
    """Chunks of derived quantities evaluated by the reference's Quantity tree (mlmc/quantity/quantity.py: arithmetic,
    ufuncs, comparisons, select, indexing, time interpolation, QArray) on a seeded 3-level storage of the SynthSimulation
    result format.  The trees are tests/zoo.py::expression_zoo built from the reference's objects; inputs are
    regenerated by tests/zoo.py::level_data(n, seed), outputs land in G8_quantity_tree.npz."""
    sys.path.insert(0, os.path.dirname(OUT))                       # tests/
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))      # repo root
    import zoo
    import mlmc.quantity.quantity as q
    from mlmc.quantity.quantity_spec import QuantitySpec
    n, seed = (60, 45, 30), 11
    fmt = zoo.result_format(QuantitySpec)
    arrs = []
    for fine, coarse in zoo.level_data(n, seed):
        arrs.append(np.stack([fine, np.zeros_like(fine) if coarse is None else coarse], axis=1))     # [N, 2, M]
    st = make_storage(arrs, [[0.5], [0.1], [0.02]], [1.0, 2.0, 3.0], fmt)
    root = q.make_root_quantity(st, fmt)
    trees = zoo.expression_zoo(root, q.Quantity)
    out = {"n": np.array(n), "seed": np.array(seed)}
    with np.errstate(all="ignore"):
        for name, tree in trees.items():
            for chunk in st.chunks():
                out["{}__L{}".format(name, chunk.level_id)] = np.asarray(tree.samples(chunk), dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "G8_quantity_tree.npz"), **out)
    print("G8", len(out) - 2, "chunks of", len(trees), "trees")


def g9_sampler_loop():
    """The reference's adaptive sampling loop (test/test_run.py:60-105): Sampler + OneProcessPool + SynthSimulation +
    Memory, `estimate_diff_vars_regression` -> `estimate_n_samples_for_target_variance` -> `process_adding_samples`
    until the estimate is met.  Recorded per round: the variances, n_estimated, the scheduled counts after the call and
    the returned flag; at the end the moments.  Harness-side accommodations: Memory._save_successful is fed through an
    object array (np.array of ragged tuples raises on NumPy >= 1.24) and the cost per sample is
    SynthSimulation.n_ops_estimate(step) instead of the pool's measured wall time (not reproducible)."""
    ruamel = types.ModuleType("ruamel")
    ruamel.yaml = types.ModuleType("ruamel.yaml")
    sys.modules.setdefault("ruamel", ruamel)
    sys.modules.setdefault("ruamel.yaml", ruamel.yaml)
    import scipy.stats as stats
    import mlmc.moments as mm
    import mlmc.quantity.quantity as q
    import mlmc.estimator as est
    from mlmc.sample_storage import Memory
    from mlmc.sampler import Sampler
    from mlmc.sampling_pool import OneProcessPool
    from mlmc.sim.synth_simulation import SynthSimulation

    class HarnessMemory(Memory):
        def _save_successful(self, samples):
            boxed = {}
            for level_id, res in samples.items():
                arr = np.empty((len(res), 2), dtype=object)
                for i, (sid, pair) in enumerate(res):
                    arr[i, 0] = sid
                    arr[i, 1] = pair
                boxed[level_id] = _Rows(arr)
            super()._save_successful(boxed)

    class _Rows:                                   # np.array(res) of the reference returns the object array as is
        def __init__(self, arr):
            self.arr = arr

        def __array__(self, dtype=None, copy=None):
            return self.arr

        def __len__(self):
            return len(self.arr)

    cases = []
    for name, step_range, n0, target_var, n_moments, loc, scale in (
            ("two_levels_test_run", [[0.1], [0.001]], [10, 10], 1e-3, 5, 0.0, 1.0),
            ("three_levels", [[0.3], [0.03], [0.003]], [200, 50, 20], 2e-5, 8, 0.0, 1.0),
            ("five_levels_shifted", [[0.5], [0.19], [0.07], [0.027], [0.01]], [300, 10], 2e-5, 6, 1.0, 2.0)):
        np.random.seed(1234)
        distr = stats.norm(loc=loc, scale=scale)
        sim = SynthSimulation(dict(distr=distr, complexity=2, nan_fraction=0))
        st = HarnessMemory()
        sampler = Sampler(sample_storage=st, sampling_pool=OneProcessPool(), sim_factory=sim, level_parameters=step_range)
        st.get_n_ops = lambda sr=step_range, sim=sim: [sim.n_ops_estimate(s[0]) for s in sr]
        true_domain = distr.ppf([0.0001, 0.9999])
        fn = mm.Legendre(n_moments, true_domain)
        sampler.set_initial_n_samples(n0)
        sampler.schedule_samples()
        sampler.ask_sampling_pool_for_samples()
        root = q.make_root_quantity(st, q_specs=sim.result_format())
        value = root['length'][1]['10'][0]
        e = est.Estimate(value, st, fn)
        rounds = []
        initial = [int(v) for v in sampler._n_scheduled_samples]
        while True:
            variances, n_ops = e.estimate_diff_vars_regression(sampler._n_scheduled_samples)
            n_est = est.estimate_n_samples_for_target_variance(target_var, variances, n_ops, n_levels=sampler.n_levels)
            done = sampler.process_adding_samples(n_est, 0, 0.1)
            rounds.append(dict(variances=np.asarray(variances).tolist(), n_ops=list(map(float, n_ops)),
                               n_estimated=[int(v) for v in n_est], n_scheduled=[int(v) for v in sampler._n_scheduled_samples],
                               n_finished=[int(v) for v in sampler.n_finished_samples], done=bool(done)))
            if done:
                break
        means, vars_ = e.estimate_moments(fn)
        cases.append(dict(name=name, level_parameters=step_range, initial=n0, initial_scheduled=initial, target_var=target_var,
                          n_moments=n_moments, loc=loc, scale=scale, domain=list(map(float, true_domain)), rounds=rounds,
                          means=np.asarray(means).tolist(), vars=np.asarray(vars_).tolist(),
                          n_collected=[int(v) for v in st.get_n_collected()]))
        print("G9", name, len(rounds), "rounds ->", cases[-1]["n_collected"])
    with open(os.path.join(OUT, "G9_sampler_loop.json"), "w") as f:
        json.dump(dict(cases=cases), f)

def g10_log_edges():
    """Samples that sit ON the edges of a log-domain (Moments(log=True, safe_eval=True), moments.py:27-39,58-73): the
    reference's keep / drop decision there depends on the last bit of np.log, which the device log() does not share.
    The grid holds the doubles around both edges of the reference's keep interval (found here by bisection over the
    reference's own ``transform``), the domain end points and their neighbours, non-positive values, inf and NaN; the
    fixture records np.log(grid) as well, so a host whose NumPy log differs from this container's in one of these points
    can be told apart from a product failure.  Also a two-level estimate_mean whose samples are drawn from that grid."""
    import mlmc.moments as mm
    import mlmc.quantity.quantity as q
    import mlmc.quantity.quantity_estimate as qe

    out = {}
    cases = (("a", (0.05, 30.0), None), ("b", (1e-3, 7.5), (-0.5, 0.75)), ("c", (2.0, 2.0000001), None),
             ("d", (1e-300, 1e300), None))
    for tag, ldom, ref in cases:
        fn = mm.Legendre(5, ldom, ref_domain=ref, log=True)

        def kept(bits, fn=fn):
            with np.errstate(all="ignore"):
                return not np.isnan(fn.transform(np.array([bits], dtype=np.uint64).view(np.float64))[0])
        top = 0x7ff0000000000000
        # any kept point to split the two searches: the geometric middle of the domain
        mid_bits = int(np.array([np.sqrt(ldom[0]) * np.sqrt(ldom[1])]).view(np.uint64)[0])
        assert kept(mid_bits)
        lo, hi = 1, mid_bits
        while lo < hi:
            m = (lo + hi) // 2
            if kept(m):
                hi = m
            else:
                lo = m + 1
        first = lo
        lo, hi = mid_bits, top - 1
        while lo < hi:
            m = (lo + hi + 1) // 2
            if kept(m):
                lo = m
            else:
                hi = m - 1
        last = lo
        edge_bits = []
        for centre in (first, last, int(np.array([ldom[0]]).view(np.uint64)[0]), int(np.array([ldom[1]]).view(np.uint64)[0])):
            edge_bits += [centre + k for k in range(-4, 5)]
        grid = np.concatenate([np.array(edge_bits, dtype=np.uint64).view(np.float64),
                               np.geomspace(ldom[0], ldom[1], 9), [0.0, -0.0, -1.0, 5e-324, 1e-320, np.inf, -np.inf, np.nan]])
        out[tag + "_ldom"] = np.array(ldom)
        out[tag + "_ref"] = np.array(ref if ref is not None else (-1.0, 1.0))
        out[tag + "_grid"] = grid
        out[tag + "_keep_interval"] = np.array([first, last], dtype=np.uint64).view(np.float64)
        with np.errstate(all="ignore"):
            out[tag + "_log"] = np.log(grid)
            out[tag + "_legendre5"] = fn.eval_all(grid)
            out[tag + "_monomial3"] = mm.Monomial(3, ldom, ref_domain=ref if ref is not None else None, log=True).eval_all(grid)
            out[tag + "_legendre5_nosafe"] = mm.Legendre(5, ldom, ref_domain=ref, log=True, safe_eval=False).eval_all(grid)
        # estimate_mean over two levels whose samples are drawn (seeded) from the grid: counts hinge on the edge decisions
        rng = np.random.default_rng(77)
        finite_pos = grid[np.isfinite(grid)]
        N = [4001, 3001]
        arrs = []
        for l, n in enumerate(N):
            a = np.empty((n, 2, 1))
            a[:, 0, 0] = rng.choice(finite_pos, size=n)
            a[:, 1, 0] = rng.choice(finite_pos, size=n) if l > 0 else 0.0
            arrs.append(a)
        spec = _scalar_spec()
        st = make_storage(arrs, [[0.5], [0.1]], [1.0, 2.0], spec)
        quantity = q.make_root_quantity(st, spec)['q'][1]['0'][0, 0]
        with np.errstate(all="ignore"):
            r = qe.estimate_mean(qe.moments(quantity, fn))
        out[tag + "_est_fine0"] = arrs[0][:, 0, 0]
        out[tag + "_est_fine1"] = arrs[1][:, 0, 0]
        out[tag + "_est_coarse1"] = arrs[1][:, 1, 0]
        out[tag + "_est_n"] = r.n_samples
        out[tag + "_est_n_rm"] = r.n_rm_samples
        out[tag + "_est_l_means"] = r.l_means
        out[tag + "_est_l_vars"] = r.l_vars
        print("G10", tag, "keep interval", out[tag + "_keep_interval"], "n", r.n_samples, "n_rm", r.n_rm_samples)
    np.savez_compressed(os.path.join(OUT, "G10_log_edges.npz"), **out)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    _install_shims()
    which = sys.argv[1:] or ["g1", "g2", "g5", "g7", "g8", "g9", "g10"]
    if "g1" in which:
        g1_basis()
    if "g2" in which:
        g2_g3_g4()
    if "g5" in which:
        g5_g6()
    if "g7" in which:
        g7_chain()
    if "g8" in which:
        g8_quantity_tree()
    if "g9" in which:
        g9_sampler_loop()
    if "g10" in which:
        g10_log_edges()
