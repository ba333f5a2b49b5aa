# Same-box A/B of two BUILDS of the library (compile-time switches):  bash tools/ab_lib.sh <variant> [sizes...]
#   python -m lsqfitgp_amd._build q0 LGP_P4_QUAD=0        (here, before gpurun: the .so travels with the snapshot)
cd $GRAFT_REPO_ROOT
V=$1; shift
SIZES="${@:-1024 2048 4096 8192 12288 16384}"
for rep in 1 2; do
  for lib in "" "$GRAFT_REPO_ROOT/lsqfitgp_amd/liblsqfitgp_hip_$V.so"; do
    echo "== library ${lib:-default}"
    for n in $SIZES; do LGP_LIBRARY=$lib timeout 100 python tools/prof_chol.py $n 6 | sort -t' ' -k6 -n | head -1; done
  done
done
