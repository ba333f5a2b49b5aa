#!/bin/bash
# fused Gram + regularisation kernel at C3 under rocprofv3 --stats, repeated: tools/gram_reg_order.sh
# (round 3: the LGP_GRAM_REG_ORDER switches this script was written for -- tile rows from both ends, no zero fill, no row sums --
#  are gone from the kernel; what they measured is in DESIGN.md section 4)
export TMPDIR=/tmp
for o in ${ORDERS:-0 1 0 1}; do
  export LGP_GRAM_REG_ORDER=$o
  out=gpurun_out/greg_$o
  rm -rf $out
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 bench.py --config c3 --steps 3 --warmup 1 --no-cpu-baseline --no-sweep > $out.json 2> $out.log
  echo "order $o: $(grep -h gram_reg_kernel $out/*/*kernel_stats.csv | head -n 1 | cut -c1-200)"
done
