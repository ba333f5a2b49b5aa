#!/bin/bash
# instruction counters of the BART Gram kernel at C5 (run on the GPU box): bash tools/pmc_bart.sh  -> gpurun_out/pmc_bart/summary.json
set -e
out=gpurun_out/pmc_bart
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES" "SQ_INSTS_VALU_TRANS SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32" "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT64 SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -- python3 tools/bench_bart_gram.py 3 > $out/p$i.log 2>&1 || echo "pass $i failed"
done
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 tools/bench_bart_gram.py 5 > $out/stats.log 2>&1
python3 - $out <<'PY'
import csv, glob, sys, json, collections
out = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob(out + '/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'gram_bart_kernel' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
res = {k: sum(v) / len(v) for k, v in acc.items()}
dur = None
for f in glob.glob(out + '/stats/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'gram_bart_kernel' in r['Name']:
            dur = float(r['AverageNs']) * 1e-9
n = 4096
pairs = n * n
res['pairs'] = pairs
res['avg_launch_s'] = dur
if 'SQ_INSTS_VALU' in res:
    res['valu_wave_instructions_per_pair'] = res['SQ_INSTS_VALU'] * 64 / pairs / 64      # wave instructions / (pairs / 64 lanes)... = per-lane instructions per pair
    res['valu_lane_instructions_per_pair'] = res['SQ_INSTS_VALU'] * 64 / pairs
    if dur:
        peak = 256 * 4 * 2.4e9 / 4                 # wave64 VALU instructions per second: one per 4 cycles per SIMD
        res['valu_wave_instr_per_s'] = res['SQ_INSTS_VALU'] / dur
        res['valu_peak_wave_instr_per_s'] = peak
        res['valu_frac_of_peak'] = res['SQ_INSTS_VALU'] / dur / peak
json.dump(res, open(out + '/summary.json', 'w'), indent=1)
print(json.dumps(res, indent=1))
PY
