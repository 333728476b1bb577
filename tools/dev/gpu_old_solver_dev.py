"""How close is mlmc_amd.tool.distribution.Distribution to the reference's multipliers / densities of G6 (*_old_*)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from mlmc_amd import _lib, Legendre
from mlmc_amd.tool import distribution as dd
from mlmc_amd.tool.simple_distribution import _solve_on_device
_lib.init(0)
g6 = np.load(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", "G6_maxent.npz"))
for name in ("norm12", "norm110", "lognorm"):
    for R in (5, 11):
        key = f"{name}_old_R{R}"
        dom = tuple(g6[key + "_domain"])
        base = Legendre(R, dom)
        ref_lam = g6[key + "_multipliers"]
        for tol in (1e-6, 1e-10):
            d = dd.Distribution(base, g6[key + "_moment_data"].copy(), domain=dom, force_decay=(True, True))
            res = d.estimate_density_minimize(tol=tol, reg_param=0.0)
            xg = g6[key + "_xgrid"]
            ref = g6[key + "_density"]
            got = d.density(xg)
            print(key, "tol", tol, "nit", res.nit, "fun_norm %.2e" % res.fun_norm, "ref nit", int(g6[key + "_nit"]), "ref fun_norm %.2e" % float(g6[key + "_fun_norm"]),
                  "| dlam rel %.2e" % (np.max(np.abs(d.multipliers - ref_lam)) / np.max(np.abs(ref_lam))),
                  "dens rel %.2e" % (np.max(np.abs(got - ref)) / np.max(ref)))
        # our gradient at the reference's multipliers (penalised functional, no stabilisation)
        lam, grad, hess, info = _solve_on_device(base, d._moment_means, d._moment_errs, dom, ref_lam.copy(), tol=1e300, max_it=1,
                                                 n_intervals=64, gauss_degree=21, stab_penalty=0.0, penalty_coef=10, decay=(True, True), prev=ref_lam.copy())
        print("    our gradient norm at the reference multipliers: %.3e  (info.grad_norm %.3e, nit %d)" % (np.linalg.norm(grad), info.grad_norm, info.nit))
