#!/bin/bash
run() { echo "== $1 | N=$2"; env $1 timeout -k 20 200 python tools/prof_chol_rhs.py $2 2>&1 | tail -6; }
run "LGP_SPLIT_MERGE_U2=0" 32768
run "LGP_X=0" 32768
run "LGP_X=0" 24576
run "LGP_SPLIT_MERGE_U2=0" 24576
