#!/usr/bin/env python3
"""Condense the raw rocprofv3 output of tools/collect_profiles.sh (gpurun_out/<round>/) into profiles/<round>_*."""
import collections
import csv
import glob
import json
import os
import sys

rnd = sys.argv[1] if len(sys.argv) > 1 else "r02"
src = os.path.join("gpurun_out", rnd)
dst = "profiles"
os.makedirs(dst, exist_ok=True)


def one(pattern):
    f = sorted(glob.glob(os.path.join(src, pattern)), key=os.path.getmtime)   # gpurun merges into the local copy: newest wins
    return f[-1] if f else None


for cfg in ("c2", "c3", "c6", "c5", "cd"):
    f = one(f"trace_{cfg}/*/*_kernel_stats.csv")
    if f:
        rows = list(csv.DictReader(open(f)))
        tag = "default_command" if cfg == "cd" else "config" + cfg[1]
        with open(os.path.join(dst, f"{rnd}_kernel_stats_{tag}.csv"), "w", newline="") as out:
            w = csv.writer(out)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
            for r in rows:
                w.writerow([r["Name"][:110], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
    summary = {}
    for kind in ("fetch", "write", "sq") + (("sq_m64",) if cfg == "c3" else ()):
        f = one(f"pmc_{kind}_{cfg}/*/*_counter_collection.csv") if kind != "sq_m64" else one("pmc_sq_m64/*/*_counter_collection.csv")
        if not f:
            continue
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        meta = {}
        for r in csv.DictReader(open(f)):
            if "mlmc::" not in r["Kernel_Name"]:
                continue
            k = r["Kernel_Name"].split("(")[0][:80]
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta[k] = dict(grid=r["Grid_Size"], wg=r["Workgroup_Size"], vgpr=r["VGPR_Count"], agpr=r["Accum_VGPR_Count"],
                           sgpr=r["SGPR_Count"], lds=r["LDS_Block_Size"])
        for k, v in agg.items():
            if kind == "sq_m64":
                if "k_moments_accum" not in k:
                    continue
                k = k + " [stand-alone moments estimates, tools/moments_only.py]"
                e = summary.setdefault(k, dict(meta[k.split(" [")[0]]))
            else:
                e = summary.setdefault(k, dict(meta[k]))
            for c, x in v.items():
                e[c + "_avg_per_dispatch"] = sum(x) / len(x)
                e["dispatches_" + kind] = len(x)
    if summary:
        sha_file = os.path.join(src, "csrc_sha.txt")
        if os.path.exists(sha_file):      # the build the counters were collected from (bench.csrc_sha on the GPU box)
            summary["_meta"] = {"csrc_sha": open(sha_file).read().strip()}
        with open(os.path.join(dst, f"{rnd}_pmc_config{cfg[1]}.json"), "w") as out:
            json.dump(summary, out, indent=1, sort_keys=True)
    for name in (f"bench_{cfg}.json",):
        p = os.path.join(src, name)
        if os.path.exists(p) and os.path.getsize(p):
            with open(p) as fi, open(os.path.join(dst, f"{rnd}_{name}"), "w") as fo:
                fo.write(fi.read())
print("profiles written:", sorted(os.listdir(dst)))
