#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel trace + PMC passes of bench.py; raw output under gpurun_out/.
# Counters are collected in their own runs (never combined with the trace domains).  The program after `--` is python3
# itself (no env / shell wrappers: the profiler's preloaded library has initialised the GPU by then).
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
R=${1:-r03}
O=gpurun_out/$R
rm -rf $O
mkdir -p $O
python3 -c "import bench; print(bench.csrc_sha())" > $O/csrc_sha.txt
# the plain bench runs come first: the profiler passes (PMC in particular) can leave the GPU in another clock mode
timeout -k 10 400 python3 bench.py > $O/bench_c3.json 2> $O/bench_c3.err
echo "bench default rc $?" > $O/progress.txt
timeout -k 10 300 python3 bench.py --config 2 --steps 200 --warmup 50 > $O/bench_c2.json 2> $O/bench_c2.err
timeout -k 10 300 python3 bench.py --config 6 > $O/bench_c6.json 2> $O/bench_c6.err
timeout -k 10 300 python3 bench.py --config 5 --steps 50 --warmup 10 > $O/bench_c5.json 2> $O/bench_c5.err
echo "plain benches done" >> $O/progress.txt
# kernel traces: the default command (BASELINE configs[2] + secondary blocks), configs[1], the tree config
# config3 = the headline workload alone (its kernel averages are not mixed with the 2e7-sample launches of the north-star block);
# "default" = the whole default command with the secondary blocks
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c3 -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/trace_c3.log 2>&1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_cd -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/trace_cd.log 2>&1
echo "trace c3 rc $?" >> $O/progress.txt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c2 -- python3 bench.py --config 2 --steps 200 --warmup 50 --no-cpu-baseline > $O/trace_c2.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c6 -- python3 bench.py --config 6 --steps 50 --warmup 10 --no-cpu-baseline > $O/trace_c6.log 2>&1
echo "traces done" >> $O/progress.txt
# PMC passes (separate runs): HBM traffic, then SQ counters
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_c3 -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > $O/pmc_fetch_c3.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_c3 -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > $O/pmc_write_c3.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq_c3 -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > $O/pmc_sq_c3.log 2>&1
echo "pmc c3 done" >> $O/progress.txt
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_c2 -- python3 bench.py --config 2 --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_fetch_c2.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_c2 -- python3 bench.py --config 2 --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_write_c2.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq_c2 -- python3 bench.py --config 2 --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_sq_c2.log 2>&1
echo "pmc c2 done" >> $O/progress.txt
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_c6 -- python3 bench.py --config 6 --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_fetch_c6.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_c6 -- python3 bench.py --config 6 --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_write_c6.log 2>&1
# the moments kernels alone (term-split kernel R = 64 mean + variance; mean-only 127 terms): their own SQ counters
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq_m64 -- python3 tools/moments_only.py > $O/pmc_sq_m64.log 2>&1
echo done >> $O/progress.txt
echo done
