cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for which in plain forced; do
  d=gpurun_out/trace_fc_$which
  rm -rf $d && mkdir -p $d
  if [ $which = forced ]; then export LGP_DIST_FORCE_COMM=1; fi
  export LGP_SPLIT_CHAIN_TAIL=0
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $d -- python3 tools/prof_dist.py c3 > $d/run.txt 2>&1 || exit 1
  f=$(find $d -name '*kernel_trace.csv' | sort | tail -1)
  python3 tools/timeline.py $f > $d/timeline.txt 2>&1
  rm -f $f
  grep "rep 3" $d/run.txt
done
