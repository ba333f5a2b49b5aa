#!/bin/bash
# N between 8192 and 16384: the chain-kernel driver for all of it (LGP_CHAIN_MAX_N=16384) against the split driver with a chain tail: tools/mid_sweep.sh
for n in 10240 12288 14336 16384; do
  echo "== N=$n chain driver"
  LGP_CHAIN_MAX_N=16384 timeout -k 10 120 python3 tools/prof_chol.py $n 4 | tail -n 2
  for t in 4096 6144 8192; do
    echo "== N=$n split + chain tail $t"
    LGP_CHAIN_MAX_N=8192 LGP_SPLIT_CHAIN_TAIL=$t timeout -k 10 120 python3 tools/prof_chol.py $n 4 | tail -n 2
  done
done
