# clocks and power while a C3-sized factorisation runs again and again (is the fp64 MFMA peak attainable in sustained operation?)
cd $GRAFT_REPO_ROOT
python tools/prof_chol.py 32768 150 > gpurun_out/clock_watch_chol.log 2>&1 &
pid=$!
sleep 14
for i in 1 2 3 4 5; do
  rocm-smi --showclocks --showpower 2>&1 | grep -i "sclk\|power (W)" | head -20
  echo "-- $(tail -1 gpurun_out/clock_watch_chol.log)"
  sleep 2
done
wait $pid
