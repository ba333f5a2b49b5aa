#!/bin/bash
# instruction counters of the fused Gram + regularisation kernel at C3 (run on the GPU box): bash tools/pmc_gram_reg.sh -> gpurun_out/pmc_gram_reg/summary.json
# (counter passes serialise kernels: the launch-per-operation driver, LGP_CHAIN=0; the Gram kernel is the same)
out=gpurun_out/pmc_gram_reg
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
export LGP_CHAIN=0
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES" "SQ_INSTS_VALU_TRANS SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -- python3 bench.py --config c3 --steps 1 --warmup 0 --no-cpu-baseline --no-sweep > $out/p$i.json 2> $out/p$i.log || echo "pass $i failed"
done
unset LGP_CHAIN
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --config c3 --steps 3 --warmup 1 --no-cpu-baseline --no-sweep > $out/stats.json 2> $out/stats.log
python3 - $out <<'PY'
import csv, glob, sys, json, collections
out = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob(out + '/p*/**/*counter_collection.csv', recursive=True):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if 'gram_reg_kernel' in r['Kernel_Name']:
            per[(r['Counter_Name'], r.get('Dispatch_Id', '0'))] += float(r['Counter_Value'])
    for (name, _), v in per.items():
        acc[name].append(v)
res = {k: sum(v) / len(v) for k, v in acc.items()}
dur = None
for f in glob.glob(out + '/stats/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'gram_reg_kernel' in r['Name']:
            dur = float(r['AverageNs']) * 1e-9
n = 32768
entries = n * (n + 1) / 2
res['entries_evaluated_lower_triangle'] = entries
res['avg_launch_s'] = dur
if 'SQ_INSTS_VALU' in res:
    res['valu_wave_instructions_per_64_entries'] = res['SQ_INSTS_VALU'] / (entries / 64)
    if dur:
        res['valu_wave_instructions_per_s'] = res['SQ_INSTS_VALU'] / dur
        res['valu_issue_peak_per_s'] = 256 * 4 * 2.4e9 / 4      # one fp64 VALU wave instruction per 4 cycles and SIMD
        res['frac_of_fp64_valu_issue_peak'] = res['valu_wave_instructions_per_s'] / res['valu_issue_peak_per_s']
res['note'] = 'rocprofv3 --pmc passes of bench.py c3 (1 step, LGP_CHAIN=0), gram_reg_kernel<double, 3, 3, false> only; duration from a --stats pass of the default run'
json.dump(res, open(out + '/summary.json', 'w'), indent=1)
print(json.dumps(res, indent=1))
PY
