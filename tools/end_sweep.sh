# the switches that shape the end of a large factorisation (the last ~14000 rows), one at a time against the defaults
cd $GRAFT_REPO_ROOT
run() { echo "== $*"; for n in 32768 16384; do env "$@" timeout 120 python tools/prof_chol.py $n 4 | sort -t' ' -k6 -n | head -1; done; }
run LGP_NOP=1
run LGP_GEMM_T64_TRAILING_MAX=5000
run LGP_GEMM_T64_TRAILING_MAX=2500
run LGP_GEMM_T64_TRAILING_MAX=20000
run LGP_THIN_LDS_KB=0
run LGP_THIN_LDS_KB=20
run LGP_SPLIT_CHAIN_TAIL=10240
run LGP_SPLIT_CHAIN_TAIL=12288
run LGP_SPLIT_CHAIN_TAIL=6144
run LGP_NOP=1
