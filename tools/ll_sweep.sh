#!/bin/bash
run() { echo "== $1 | N=$2"; env $1 timeout -k 10 120 python tools/prof_chol.py $2 4 2>&1 | tail -2; }
for N in 8192 4096 2048 12288 16384; do
  run "LGP_CHAIN_U2_WHOLE=1" $N
  run "LGP_X=0" $N
  run "LGP_CHAIN_U2_AFTER_U1_MAX_N=0" $N
  run "LGP_CHAIN_U2_AFTER_U1_MAX_N=16384" $N
done
