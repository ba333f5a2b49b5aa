#!/bin/bash
# kernel trace of the Cholesky at N (default 8192) + the chain kernel's task trace:  bash tools/trace_8192.sh [N]
N=${1:-8192}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/trace_$N && mkdir -p gpurun_out/trace_$N
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_$N -- python3 tools/prof_chol.py $N 3 > gpurun_out/trace_$N/run.txt 2>&1
f=$(find gpurun_out/trace_$N -name '*kernel_trace.csv' | sort | tail -1)
python3 tools/timeline.py $f > gpurun_out/trace_$N/timeline.txt 2>&1
rm -f $f; find gpurun_out/trace_$N -name '*.csv' -size +20M -delete
LGP_CHAIN_TRACE=gpurun_out/trace_$N/chain.bin timeout -k 10 100 python3 tools/prof_chol.py $N 3 > gpurun_out/trace_$N/run2.txt 2>&1
python3 tools/chain_trace.py gpurun_out/trace_$N/chain.bin > gpurun_out/trace_$N/chain.txt 2>&1 || true
tail -3 gpurun_out/trace_$N/run.txt
