#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel trace + PMC passes of bench.py; raw output under gpurun_out/.
# Counters are collected in their own runs (never combined with the trace domains).
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
R=${1:-r01}
O=gpurun_out/$R
rm -rf $O
mkdir -p $O
# the plain bench runs come first: the profiler passes (PMC in particular) can leave the GPU in another clock mode
timeout -k 10 300 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err
timeout -k 10 300 python bench.py --config 6 > $O/bench_c6.json 2> $O/bench_c6.err
timeout -k 10 300 python bench.py --config 3 --steps 20 --warmup 5 > $O/bench_c3.json 2> $O/bench_c3.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c2 -- python bench.py --steps 200 --warmup 50 --no-cpu-baseline > $O/trace_c2.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c3 -- python bench.py --config 3 --steps 10 --warmup 3 --no-cpu-baseline > $O/trace_c3.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_c2 -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_fetch_c2.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_c2 -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_write_c2.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq_c2 -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_sq_c2.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_c3 -- python bench.py --config 3 --steps 1 --warmup 1 --no-cpu-baseline > $O/pmc_fetch_c3.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq_c3 -- python bench.py --config 3 --steps 1 --warmup 1 --no-cpu-baseline > $O/pmc_sq_c3.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c6 -- python bench.py --config 6 --steps 50 --warmup 10 --no-cpu-baseline > $O/trace_c6.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_c6 -- python bench.py --config 6 --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_fetch_c6.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_c6 -- python bench.py --config 6 --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_write_c6.log 2>&1
echo done
