#!/bin/bash
# kernel trace of the C5 step loop (tools/prof_c5.py): per-kernel totals and the timeline of the last step
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/trace_c5 && mkdir -p gpurun_out/trace_c5
timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d gpurun_out/trace_c5 -- python3 tools/prof_c5.py > gpurun_out/trace_c5/run.txt 2>&1
f=$(find gpurun_out/trace_c5 -name '*kernel_trace.csv' | sort | tail -1)
python3 - $f > gpurun_out/trace_c5/timeline.txt <<'PY'
import csv, sys
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
# the last step starts at the last gram_bart_kernel with a big grid: take the last 3 bart launches
idx = [i for i, r in enumerate(rows) if 'gram_bart_kernel' in r[2]]
i0 = idx[-3]
t0 = rows[i0][0]
busy = 0; last = t0
for s, e, n in rows[i0:]:
    n = n.replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '')[:80]
    print(f'{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} gap {(s - last) / 1e3:7.1f}  {n}')
    busy += (e - s) / 1e3; last = max(last, e)
print('busy', busy, 'span', (last - t0) / 1e3)
PY
cp $(find gpurun_out/trace_c5 -name '*kernel_stats.csv' | tail -1) gpurun_out/trace_c5/kernel_stats.csv
find gpurun_out/trace_c5 -name '*trace.csv' -delete
head -3 gpurun_out/trace_c5/run.txt
