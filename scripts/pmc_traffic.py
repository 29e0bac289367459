"""Turn gpurun_out/profiles_run/ (scripts/collect_profiles.sh) into the committed files under profiles/:
r01_bench.json, r01_bench_kernel_stats.csv and dominant_kernel_pmc.json (HBM bytes per launch of the dominant
kernel family, corrected as MI355X_MICROARCH.md prescribes: bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024)."""
import csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
run = os.path.join(ROOT, 'gpurun_out', 'profiles_run')
tag = sys.argv[1] if len(sys.argv) > 1 else 'r01'
frags = ('conv_pf_kernelIDF16bLi9ELi2E', '11conv_kernelIDF16bLi9ELi2E', 'conv_pp_kernel')          # == bench.DOMINANT_SYMBOLS

line = [l for l in open(os.path.join(run, 'bench.json')) if l.startswith('{')][-1]
open(os.path.join(ROOT, 'profiles', tag + '_bench.json'), 'w').write(line)
stats = glob.glob(os.path.join(run, 'stats', '*kernel_stats.csv'))[0]
shutil.copy(stats, os.path.join(ROOT, 'profiles', tag + '_bench_kernel_stats.csv'))
rows = list(csv.DictReader(open(stats)))
dom = [r for r in rows if any(f in r['Name'] for f in frags)]
calls = sum(int(r['Calls']) for r in dom)
avg_us = sum(float(r['TotalDurationNs']) for r in dom) / calls / 1e3
share = sum(float(r['Percentage']) for r in dom)

vals = {'FETCH_SIZE': [], 'WRITE_SIZE': []}
for f in glob.glob(os.path.join(run, 'pmc', 'pmc_*', '*counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        if any(fr in r['Kernel_Name'] for fr in frags) and r['Counter_Name'] in vals:
            vals[r['Counter_Name']].append(float(r['Counter_Value']))
fk = sum(vals['FETCH_SIZE']) / len(vals['FETCH_SIZE'])
wk = sum(vals['WRITE_SIZE']) / len(vals['WRITE_SIZE'])
out = {
    'kernel': 'conv_kernel<bf16,9,2> (conv_pf_kernel<bf16,9,2,*> + conv_kernel<bf16,9,2> + conv_pp_kernel)',
    'launches_profiled': len(vals['FETCH_SIZE']),
    'fetch_size_kb_per_launch': fk, 'write_size_kb_per_launch': wk,
    'traffic_bytes_per_launch': (2 * fk + wk) * 1024,
    'rocprof_avg_launch_us': avg_us, 'rocprof_launches': calls, 'rocprof_share_percent': share,
    'correction': 'HBM bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (MI355X_MICROARCH.md: FETCH_SIZE reports half of 16-B/lane '
                  'streaming reads on gfx950; counters in KB)',
}
json.dump(out, open(os.path.join(ROOT, 'profiles', 'dominant_kernel_pmc.json'), 'w'), indent=1)
print(json.dumps(out, indent=1))
print(line)
