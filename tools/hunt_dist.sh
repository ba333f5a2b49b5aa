# the 4-rank case of tests/test_gpu_dist.py again and again under pytest (as the suite runs it), until it fails or the time is up
cd $GRAFT_REPO_ROOT
end=$(( $(date +%s) + ${1:-600} ))
i=0
while [ $(date +%s) -lt $end ]; do
  i=$((i+1))
  if ! python -m pytest tests/test_gpu_dist.py -x -q -m gpu -k "4100-128-4" > gpurun_out/hunt_last.log 2>&1; then
    cp gpurun_out/hunt_last.log gpurun_out/hunt_FAILED_$i.log
    echo "iteration $i FAILED" >> gpurun_out/hunt.log
    exit 0
  fi
  echo "iteration $i ok $(tail -1 gpurun_out/hunt_last.log)" >> gpurun_out/hunt.log
done
echo "done: $i iterations, no failure" >> gpurun_out/hunt.log
