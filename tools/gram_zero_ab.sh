#!/bin/bash
# The fused Gram kernel with and without the stores of zeros above the diagonal (LGP_FACTOR_LOWER_ONLY), same box:
#   bash tools/gram_zero_ab.sh   -> gpurun_out/gram_zero_ab.txt (kernel durations from rocprofv3 --kernel-trace --stats)
export TMPDIR=/tmp
out=gpurun_out/gram_zero_ab; rm -rf $out; mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/lower -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sweep > $out/lower.json 2> $out/lower.log || exit 1
export LGP_BENCH_FULL_FACTOR=1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/full -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sweep > $out/full.json 2> $out/full.log || exit 1
for v in lower full; do
  f=$(find $out/$v -name '*kernel_stats.csv' | head -1)
  echo "== $v"; grep -i "gram_reg\|Name" $f | cut -c1-200
done > gpurun_out/gram_zero_ab.txt
cat gpurun_out/gram_zero_ab.txt
