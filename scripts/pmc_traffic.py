"""Turn gpurun_out/profiles_run/ (scripts/collect_profiles.sh) into the committed files under profiles/:
<tag>_bench.json, <tag>_bench_kernel_stats.csv and dominant_kernel_pmc.json -- per kernel family of bench.py (bench.FAMILIES:
the fused small-channel backward, the 64-wide 3x3 convs, the stand-alone weight gradients): rocprofv3's average launch
duration and the HBM bytes per launch from the PMC passes, corrected as MI355X_MICROARCH.md prescribes:
bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024.  One weight-gradient launch = one rd_wgrad call = its MFMA kernel + its
split-reduction kernel: times and bytes of both are summed and divided by the number of MFMA-kernel launches."""
import csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
run = os.path.join(ROOT, 'gpurun_out', 'profiles_run')
tag = sys.argv[1] if len(sys.argv) > 1 else 'r01'
import importlib.util
_spec = importlib.util.spec_from_file_location('rd_bench', os.path.join(ROOT, 'bench.py'))
_bench = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_bench)
# the family table of bench.py ('count' = which symbols count as one launch: a weight-gradient launch = MFMA kernel + its reduce)
FAMILIES = {fam: dict(frags=v['symbols'], count=v['count'], name=' + '.join(v['symbols'])) for fam, v in _bench.FAMILIES.items()}

line = [l for l in open(os.path.join(run, 'bench.json')) if l.startswith('{')][-1]
open(os.path.join(ROOT, 'profiles', tag + '_bench.json'), 'w').write(line)
stats = glob.glob(os.path.join(run, 'stats', '*kernel_stats.csv'))[0]
shutil.copy(stats, os.path.join(ROOT, 'profiles', tag + '_bench_kernel_stats.csv'))
rows = list(csv.DictReader(open(stats)))
total_ns = sum(float(r['TotalDurationNs']) for r in rows)
pmc_rows = []
for f in glob.glob(os.path.join(run, 'pmc', 'pmc_*', '*counter_collection.csv')):
    pmc_rows += [r for r in csv.DictReader(open(f)) if r['Counter_Name'] in ('FETCH_SIZE', 'WRITE_SIZE')]

out = {'correction': 'HBM bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (MI355X_MICROARCH.md: FETCH_SIZE reports half of 16-B/lane '
                     'streaming reads on gfx950; counters in KB); FETCH_SIZE and WRITE_SIZE collected in separate passes'}
for fam, spec in FAMILIES.items():
    member = lambda n: any(f in n for f in spec['frags'])
    counted = lambda n: any(f in n for f in (spec['count'] or spec['frags']))
    dom = [r for r in rows if member(r['Name'])]
    calls = sum(int(r['Calls']) for r in dom if counted(r['Name']))
    tot = sum(float(r['TotalDurationNs']) for r in dom)
    kb = {'FETCH_SIZE': 0.0, 'WRITE_SIZE': 0.0}
    nl = {'FETCH_SIZE': 0, 'WRITE_SIZE': 0}
    for r in pmc_rows:
        if member(r['Kernel_Name']):
            kb[r['Counter_Name']] += float(r['Counter_Value'])
            if counted(r['Kernel_Name']):
                nl[r['Counter_Name']] += 1
    fk, wk = kb['FETCH_SIZE'] / max(nl['FETCH_SIZE'], 1), kb['WRITE_SIZE'] / max(nl['WRITE_SIZE'], 1)
    out[fam] = {
        'family': fam, 'kernels': spec['name'], 'launches_profiled': nl['FETCH_SIZE'],
        'fetch_size_kb_per_launch': fk, 'write_size_kb_per_launch': wk, 'traffic_bytes_per_launch': (2 * fk + wk) * 1024,
        'rocprof_avg_launch_us': tot / calls / 1e3, 'rocprof_launches': calls, 'rocprof_share_percent': 100.0 * tot / total_ns,
    }
# ---- every kernel of a step: HBM bytes of all dispatches of the library's kernels (torch's allocation fills excluded: they run once,
# before the first step) divided by the number of steps the profiled command ran (= dispatches of the once-per-step Adam kernel)
ours = lambda n: not n.startswith('_ZN2at') and 'at::native' not in n
nsteps = {c: sum(1 for r in pmc_rows if r['Counter_Name'] == c and 'adam_update_kernel' in r['Kernel_Name']) for c in ('FETCH_SIZE', 'WRITE_SIZE')}
tot_kb = {c: sum(float(r['Counter_Value']) for r in pmc_rows if r['Counter_Name'] == c and ours(r['Kernel_Name'])) for c in ('FETCH_SIZE', 'WRITE_SIZE')}
if min(nsteps.values()) > 0:
    fk, wk = tot_kb['FETCH_SIZE'] / nsteps['FETCH_SIZE'], tot_kb['WRITE_SIZE'] / nsteps['WRITE_SIZE']
    bl = json.loads(line)
    per_kernel = {}
    for r in pmc_rows:
        if ours(r['Kernel_Name']):
            k = r['Kernel_Name'].split('(')[0][:60]
            per_kernel.setdefault(k, {'FETCH_SIZE': 0.0, 'WRITE_SIZE': 0.0})[r['Counter_Name']] += float(r['Counter_Value'])
    top = sorted(per_kernel.items(), key=lambda kv: -(2 * kv[1]['FETCH_SIZE'] / nsteps['FETCH_SIZE'] + kv[1]['WRITE_SIZE'] / nsteps['WRITE_SIZE']))[:12]
    out['step'] = {'steps_profiled': nsteps['FETCH_SIZE'], 'fetch_size_kb_per_step': fk, 'write_size_kb_per_step': wk,
                   'traffic_bytes_per_step': (2 * fk + wk) * 1024, 'size': 400, 'dtype': bl.get('dtype'), 'batch': bl['config']['global_batch'] // bl['n_gpus'],
                   'top_kernels_mb_per_step': {k: round((2 * v['FETCH_SIZE'] / nsteps['FETCH_SIZE'] + v['WRITE_SIZE'] / nsteps['WRITE_SIZE']) / 1024, 1) for k, v in top}}
out['collected'] = 'rocprofv3 PMC passes of scripts/collect_profiles.sh, tag %s' % tag
json.dump(out, open(os.path.join(ROOT, 'profiles', 'dominant_kernel_pmc.json'), 'w'), indent=1)
print(json.dumps(out, indent=1))
# the committed bench line carries the traffic figures of THIS collection (bench.py reads them from the json that was committed when it
# ran, i.e. the previous collection's): rewrite the fields that are copies of the json, nothing measured inside the bench run
bl = json.loads(line)
for key, r in bl.items():
    if isinstance(r, dict) and r.get('family') in out and 'traffic' in r:
        r['traffic'] = int(out[r['family']]['traffic_bytes_per_launch'])
        r['traffic_over_algorithmic'] = round(r['traffic'] / r['avg_algorithmic_bytes'], 3)
if 'step' in out and 'roofline_step' in bl:
    rs = bl['roofline_step']
    rs['traffic'] = int(out['step']['traffic_bytes_per_step'])
    rs['traffic_over_algorithmic'] = round(rs['traffic'] / rs['algorithmic_bytes_per_step'], 3)
    rs['traffic_gbs'] = round(rs['traffic'] / (bl['ms_per_step'] * 1e-3) / 1e9, 1)
line = json.dumps(bl) + '\n'
open(os.path.join(ROOT, 'profiles', tag + '_bench.json'), 'w').write(line)
print('shares of the summed kernel time:', {f: round(out[f]['rocprof_share_percent'], 1) for f in FAMILIES})
print(line)
