"""Summarise the rocprofv3 passes of tools/profile_bench.sh into small files for profiles/ (run on the GPU box;
the raw traces stay in gpurun_out/):  python3 tools/profile_summarise.py <dir> <config> <tag>"""
import csv, glob, json, os, sys, collections

out, cfg, tag = sys.argv[1:4]
dst = os.path.join(out, 'summary')
os.makedirs(dst, exist_ok=True)


def one(pattern):
    files = glob.glob(os.path.join(out, pattern), recursive=True)
    return files[0] if files else None


# pass 1: kernel stats (already a summary): keep name, calls, total/avg/min/max ns, percentage
stats = one('stats/**/*kernel_stats.csv')
rows = list(csv.DictReader(open(stats)))
with open(os.path.join(dst, f'{tag}_{cfg}_kernel_stats.csv'), 'w', newline='') as f:
    w = csv.writer(f)
    w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs'])
    for r in rows:
        w.writerow([r['Name'], r['Calls'], r['TotalDurationNs'], r['AverageNs'], r['Percentage'], r['MinNs'], r['MaxNs']])

# durations of the trailing-update launches from the kernel trace: the big NT lower gemm launches on stream s0 are
# the ones with grid >= 1024 workgroups of the <double,false,true,128,...> instantiation
trace = one('stats/**/*kernel_trace.csv')
tr = list(csv.DictReader(open(trace)))
def is_nt(name):
    return 'gemm_kernel<double, false, true, 128' in name or 'gemm_kernelIdLb0ELb1ELi128' in name
def is_trailing_128(name):   # the TAG = 1 instantiation (lgp_potrf_impl's trailing update)
    return 'gemm_kernel<double, false, true, 128, 256, 1>' in name or 'gemm_kernelIdLb0ELb1ELi128ELi256ELi1E' in name
def is_small_path(name):     # configurations that never launch it (C2: N = 8192): their updates are the 64-tile NT instantiation, beside the chain kernel
    return 'gemm_kernel<double, false, true, 64' in name or 'gemm_kernelIdLb0ELb1ELi64' in name or 'chain_kernel' in name
is_trailing = is_trailing_128
bench = json.loads(open(os.path.join(out, 'bench_stats.json')).read().strip().splitlines()[-1])
n_launch = bench['roofline']['launches']
nt = [int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in tr if is_trailing(r['Kernel_Name'])]
summary = {'bench_line_under_rocprof': bench}
matched = 'gemm_kernel<double, false, true, 128, 256, 1> (the trailing updates on 128-tiles)'
if not nt:       # no 128-tile trailing update in this configuration: the counters below are those of its 64-tile updates + the chain kernel
    is_trailing = is_small_path
    matched = 'gemm_kernel<double, false, true, 64, ...> + chain_kernel (this configuration launches no 128-tile trailing update)'
    nt = [int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in tr if is_trailing(r['Kernel_Name'])]
summary['kernels_matched'] = matched
if nt:
    summary['trailing_update_launches_in_trace'] = len(nt)      # warmup + timed steps
    summary['trailing_update_avg_ms_in_trace'] = sum(nt) / len(nt) / 1e6
    summary['trailing_update_avg_ms_bench_hip_events'] = bench['roofline']['avg_launch_ms']
with open(os.path.join(dst, f'{tag}_{cfg}_bench_under_rocprof.json'), 'w') as f:
    json.dump(summary, f, indent=1)


def counter(passdir, name):
    """ per-dispatch sums of one counter for the trailing-update launches and for every NT gemm """
    cc = one(f'{passdir}/**/*counter_collection.csv')
    per = collections.OrderedDict()
    for r in csv.DictReader(open(cc)):
        if r['Counter_Name'] != name:
            continue
        key = r['Dispatch_Id']
        per.setdefault(key, [r['Kernel_Name'], int(r['Grid_Size']), 0.0])
        per[key][2] += float(r['Counter_Value'])
    return list(per.values())


def trailing(disp):
    return [d for d in disp if is_trailing(d[0])]


fetch = counter('fetch', 'FETCH_SIZE')
write = counter('write', 'WRITE_SIZE')
tf, tw = trailing(fetch), trailing(write)
b1 = json.loads(open(os.path.join(out, 'bench_fetch.json')).read().strip().splitlines()[-1])
launches = b1['roofline']['launches']
alg = None
res = {
    'note': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, 1 step of bench.py %s); units KB; gfx950 '
            'correction: FETCH_SIZE x2 for wide coalesced reads (MI355X_MICROARCH.md HBM section)' % cfg,
    'kernels_matched': matched,
    'trailing_update_launches_bench': launches,
    'trailing_update_launches_matched': len(tf),
    'fetch_size_kb_sum': sum(d[2] for d in tf),
    'write_size_kb_sum': sum(d[2] for d in tw),
    'all_gemm_nt': {'fetch_kb': sum(d[2] for d in fetch if is_nt(d[0])), 'write_kb': sum(d[2] for d in write if is_nt(d[0])),
                    'launches': sum(1 for d in fetch if is_nt(d[0]))},
}
if tf:
    res['hbm_read_bytes_per_launch_corrected'] = 2 * 1024 * res['fetch_size_kb_sum'] / len(tf)
    res['hbm_write_bytes_per_launch'] = 1024 * res['write_size_kb_sum'] / max(len(tw), 1)
    # algorithmic bytes of an average launch: read + write the lower triangle of C once, read the panel once
    if b1['roofline']['avg_launch_gflop']:
        res['algorithmic_bytes_per_launch_avg'] = b1['roofline']['avg_launch_gflop'] * 1e9 / (2.0 * 1024) * 8 * 2
        res['traffic_over_algorithmic'] = (res['hbm_read_bytes_per_launch_corrected'] + res['hbm_write_bytes_per_launch']) / res['algorithmic_bytes_per_launch_avg']
with open(os.path.join(dst, f'{tag}_{cfg}_pmc_traffic.json'), 'w') as f:
    json.dump(res, f, indent=1)
print(json.dumps(res, indent=1))

# pass 4: matrix-core utilisation of the trailing-update launches (own pass: SQ counters + kernel trace only)
try:
    names = ['SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_BUSY_CU_CYCLES', 'SQ_INSTS_MFMA', 'SQ_INSTS_VALU']
    sums = {nm: sum(d[2] for d in trailing(counter('mfma', nm))) for nm in names}
    mtrace = list(csv.DictReader(open(one('mfma/**/*kernel_trace.csv'))))
    dur_ns = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in mtrace if is_trailing(r['Kernel_Name']))
    nl = sum(1 for r in mtrace if is_trailing(r['Kernel_Name']))
    util = {
        'note': 'rocprofv3 --kernel-trace --pmc ' + ' '.join(names) + ' (1 step of bench.py %s), sums over the trailing-update launches (tag-1 gemm)' % cfg,
        'kernels_matched': matched, 'launches': nl, 'duration_ms_sum': dur_ns / 1e6, **sums,
        # an fp64 16x16x4 MFMA occupies its SIMD for 64 cycles (tools/f64_rate_probe.hip); 1024 SIMDs; the counter sums all SEs/XCDs
        'mfma_pipe_busy_frac_from_insts': sums['SQ_INSTS_MFMA'] * 64 / (dur_ns * 1e-9 * 2.4e9 * 1024) if dur_ns else None,
        'mfma_busy_cycles_per_mfma_inst': sums['SQ_VALU_MFMA_BUSY_CYCLES'] / sums['SQ_INSTS_MFMA'] if sums['SQ_INSTS_MFMA'] else None,
        'mfma_busy_over_cu_busy': sums['SQ_VALU_MFMA_BUSY_CYCLES'] / sums['SQ_BUSY_CU_CYCLES'] if sums['SQ_BUSY_CU_CYCLES'] else None,
        'valu_insts_per_mfma_inst': sums['SQ_INSTS_VALU'] / sums['SQ_INSTS_MFMA'] if sums['SQ_INSTS_MFMA'] else None,
    }
    with open(os.path.join(dst, f'{tag}_{cfg}_mfma_util.json'), 'w') as f:
        json.dump(util, f, indent=1)
    print(json.dumps(util, indent=1))
except Exception as exc:
    print('mfma utilisation pass not summarised:', repr(exc))
for r in rows[:12]:
    print(r['Name'][:90], r['Calls'], r['TotalDurationNs'], r['AverageNs'], r['Percentage'])
