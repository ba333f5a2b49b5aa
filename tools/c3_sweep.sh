#!/bin/bash
run() { echo "== $1 | N=$2"; env $1 timeout -k 10 200 python tools/prof_chol.py $2 3 2>&1 | tail -2; }
for N in 32768; do
  run "LGP_X=0" $N
  run "LGP_U2_T128_MIN_ROWS=12288" $N
  run "LGP_U2_T128_MIN_ROWS=4096" $N
  run "LGP_GEMM_T64_TRAILING_MAX=6000" $N
  run "LGP_GEMM_T64_TRAILING_MAX=14000" $N
  run "LGP_U2_T128_MIN_ROWS=12288 LGP_GEMM_T64_TRAILING_MAX=6000" $N
done
