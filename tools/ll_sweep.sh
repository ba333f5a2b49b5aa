#!/bin/bash
run() { echo "== $1 | N=$2"; env $1 timeout -k 10 120 python tools/prof_chol.py $2 4 2>&1 | tail -2; }
for N in 16384 12288; do
  run "LGP_X=0" $N
  run "LGP_CHAIN_WORKERS=8" $N
  run "LGP_CHAIN_WORKERS=12" $N
  run "LGP_CHAIN_WORKERS=24" $N
  run "LGP_CHAIN_WORKERS=8 LGP_CHAIN_HOG_MAX_N=16384" $N
  run "LGP_CHAIN_WORKERS=12 LGP_CHAIN_HOG_MAX_N=16384" $N
done
