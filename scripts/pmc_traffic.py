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

bl0 = json.loads(line)
out = _bench.pmc_aggregate(pmc_rows, bl0['config']['global_batch'] // bl0['n_gpus'], bl0.get('dtype'), 400)
for fam, spec in FAMILIES.items():          # + the profiler's own launch durations and kernel-time shares of the --stats run
    member = lambda n: any(f in n for f in spec['frags'])
    counted = lambda n: any(f in n for f in (spec['count'] or spec['frags']))
    dom = [r for r in rows if member(r['Name'])]
    calls = sum(int(r['Calls']) for r in dom if counted(r['Name']))
    tot = sum(float(r['TotalDurationNs']) for r in dom)
    if fam in out and calls:
        out[fam].update(rocprof_avg_launch_us=tot / calls / 1e3, rocprof_launches=calls, rocprof_share_percent=100.0 * tot / total_ns)
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
