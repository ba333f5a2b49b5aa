#!/bin/bash
# the split driver handing its tail to the chain-kernel driver (LGP_SPLIT_CHAIN_TAIL rows; 0 = never): tools/tail_sweep.sh
for n in 32768 20480; do
  for t in 0 4096 8192 12288 16384; do
    echo "== N=$n LGP_SPLIT_CHAIN_TAIL=$t"
    LGP_SPLIT_CHAIN_TAIL=$t timeout -k 10 120 python3 tools/prof_chol.py $n 4 | tail -n 2
  done
done
