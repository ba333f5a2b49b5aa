#!/bin/bash
# Round evidence for bench.py (run on the GPU box):  bash tools/profile_bench.sh [config] [round-tag]
#   pass 1: rocprofv3 --kernel-trace --stats   -> per-kernel durations of the SAME command bench.py runs
#   pass 2: rocprofv3 --pmc FETCH_SIZE         -> L2->fabric read traffic per kernel launch
#   pass 3: rocprofv3 --pmc WRITE_SIZE         -> write traffic
#   pass 4: rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU  -> matrix-core utilisation
# (counters in their own passes, never combined with system traces).  tools/profile_summarise.py turns the three
# outputs into profiles/<tag>_<config>_{kernel_stats.csv,bench_under_rocprof.json,pmc_traffic.json}.
set -e
cfg=${1:-c3}
tag=${2:-r02}
out=gpurun_out/prof_$cfg
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-sweep > $out/bench_stats.json 2> $out/stats.log
# (counter passes serialise the kernels of a process: the chain kernel, whose workers wait for kernels of other streams, cannot
#  run under them -- the counters are collected on the launch-per-operation driver, LGP_CHAIN=0; the trailing update is the same kernel)
export LGP_CHAIN=0
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 bench.py --config $cfg --steps 1 --warmup 0 --no-cpu-baseline --no-sweep > $out/bench_fetch.json 2> $out/fetch.log
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 bench.py --config $cfg --steps 1 --warmup 0 --no-cpu-baseline --no-sweep > $out/bench_write.json 2> $out/write.log
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU --output-format csv -d $out/mfma -- python3 bench.py --config $cfg --steps 1 --warmup 0 --no-cpu-baseline --no-sweep > $out/bench_mfma.json 2> $out/mfma.log
python3 tools/profile_summarise.py $out $cfg $tag
