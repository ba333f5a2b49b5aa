#!/bin/bash
# the fused Gram + regularisation kernel inside bench.py (C3 and C2) under rocprofv3 --kernel-trace --stats:  bash tools/gram_reg_time.sh
export TMPDIR=/tmp
for cfg in c3 c2; do
  out=gpurun_out/greg_$cfg; rm -rf $out; mkdir -p $out
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-sweep > $out/bench.json 2> $out/log.txt || exit 1
  f=$(find $out -name '*kernel_stats.csv' | head -1)
  test -n "$f" && echo "$cfg: $(grep -h gram_reg_kernel $f | head -n 1 | sed 's/.*)",//' | cut -d, -f1-3)  (calls, total ns, average ns)"
done
