#!/usr/bin/env python3
"""Kernel micro-benchmark / same-box A/B driver (run on the GPU box through gpurun).

    python tools/kbench.py CASE [CASE ...] [--libs name=path[@VAR=VAL],...] [--rounds 3] [--n 10000000]

CASE = mode:R[:basis[:levels[:flags]]]   mode in {mom, cov};  basis in {leg, mono, four, spline};  levels = number of levels
       (level 0 has no coarse samples; "0" = one level-0 launch alone, "p" = one pair level alone);  flags: m = mean only
       e.g.  cov:64:leg:5   cov:64:leg:5:m   mom:127:leg:5   cov:128:spline:1:m
Prints the HIP-event kernel time per estimate (mlmc_accum_kernel_time) and the wall time per estimate, best of `rounds`
child processes per library.  Libraries alternate (MLMC_HIP_LIB) so that the variants see the same box in the same state;
"path@VAR=VAL" runs the same library under an environment switch."""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys, os, time
sys.path.insert(0, %r)
import torch
from mlmc_amd import _lib, Legendre, Monomial, Fourier, Spline
from mlmc_amd.engine import LevelAccumulator
_lib.init(0, _lib.FLAG_TIMING)
n = int(os.environ["KB_N"])
g = torch.Generator(device="cuda"); g.manual_seed(1)
steps = [0.5, 0.19, 0.07, 0.027, 0.01, 0.004, 0.0015, 0.0006]
levels = []
for l in range(8):
    x = torch.randn(n, dtype=torch.float64, device="cuda", generator=g)
    root = torch.sqrt(1e-4 + x.abs())
    levels.append(((x + steps[l] * root).contiguous(), (x + steps[l - 1] * root).contiguous() if l else None))
torch.cuda.synchronize()
for case in os.environ["KB_CASES"].split(","):
    parts = case.split(":") + ["", "", ""]
    mode, R, basis, lv, flags = parts[0], int(parts[1]), parts[2] or "leg", parts[3] or "5", parts[4]
    cls = {"leg": Legendre, "mono": Monomial, "four": Fourier, "spline": Spline}[basis]
    dom = (-3.719, 3.719)
    fn = cls(R, dom)
    if lv == "0":
        chunks = [(0, levels[0][0], None)]
    elif lv == "p":
        chunks = [(1, levels[1][0], levels[1][1])]
    else:
        chunks = [(l, levels[l][0], levels[l][1]) for l in range(int(lv))]
    L = max(c[0] for c in chunks) + 1
    acc = LevelAccumulator(fn, L, LevelAccumulator.MOMENTS if mode == "mom" else LevelAccumulator.COV, mean_only="m" in flags)
    reps = int(os.environ.get("KB_REPS", "10"))
    for it in range(4):
        acc.estimate(chunks, reduce=False)
    acc.kernel_time()
    acc.kernel_flops()
    acc.aux_kernel_time()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(reps):
        acc.estimate(chunks, reduce=False)
    wall = (time.perf_counter() - t0) / reps
    ms, launches, nb = acc.kernel_time()
    fl = acc.kernel_flops()
    aux = acc.aux_kernel_time()[0]
    print("RES %%s %%.5f %%.5f %%d %%.4f %%.5f" %% (case, ms / reps, 1e3 * wall, launches // reps, fl / reps / (ms / reps * 1e-3) / 1e12 if ms else 0.0, aux / reps), flush=True)
    acc.close()
''' % ROOT


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("cases", nargs="+")
    ap.add_argument("--libs", default="")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--n", type=int, default=10_000_000)
    ap.add_argument("--reps", type=int, default=10)
    args = ap.parse_args()
    libs = {"lib": os.path.join(ROOT, "mlmc_amd", "libmlmc_hip.so")}
    if args.libs:
        libs = {kv.split("=", 1)[0]: kv.split("=", 1)[1] for kv in args.libs.split(",")}
    res = {name: {} for name in libs}
    for rnd in range(args.rounds):
        for name, lib in libs.items():
            extra = {}
            if "@" in lib:
                lib, kv = lib.split("@", 1)
                extra = dict([kv.split("=", 1)])
            lib = lib if os.path.isabs(lib) else os.path.join(ROOT, lib)
            env = dict(os.environ, MLMC_HIP_LIB=lib, KB_CASES=",".join(args.cases), KB_N=str(args.n), KB_REPS=str(args.reps), **extra)
            out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=900)
            if out.returncode != 0:
                print(name, "FAILED", out.stderr[-800:], flush=True)
                continue
            for line in out.stdout.splitlines():
                if line.startswith("RES "):
                    _, case, k_ms, w_ms, launches, tf, aux = line.split()
                    res[name].setdefault(case, []).append((float(k_ms), float(w_ms), int(launches), float(tf), float(aux)))
            print("round", rnd, name, "done", flush=True)
    print("%-24s %-10s %12s %12s %9s %12s %10s" % ("case", "lib", "kernel ms", "wall ms", "launches", "MFMA TF/s", "aux ms"))
    for case in args.cases:
        for name in libs:
            v = res[name].get(case)
            if v:
                best = min(v)
                print("%-24s %-10s %12.4f %12.4f %9d %12.2f %10.4f" % (case, name, best[0], min(x[1] for x in v), best[2], max(x[3] for x in v), best[4]))


if __name__ == "__main__":
    main()
