#!/bin/bash
# kernel traces of the sharded path with one rank and of the single-GPU driver at C3's size, side by side:  bash tools/trace_dist.sh
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for which in dist single; do
  d=gpurun_out/trace_$which
  rm -rf $d && mkdir -p $d
  if [ $which = dist ]; then cmd="tools/prof_dist.py c3"; else cmd="bench.py --config c3 --no-sweep --no-cpu-baseline --steps 4 --warmup 1"; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $d -- python3 $cmd > $d/run.txt 2>&1 || exit 1
  f=$(find $d -name '*kernel_trace.csv' | sort | tail -1)
  python3 tools/timeline.py $f > $d/timeline.txt 2>&1
  rm -f $f; find $d -name '*.csv' -size +20M -delete
  tail -2 $d/run.txt
done
