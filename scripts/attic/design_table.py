"""Prints DESIGN.md section 6's family table from a committed bench line (default profiles/r04_bench.json) and the kernel-time shares of
the matching rocprofv3 statistics (scripts/pmc_traffic.py prints them), so that the document and the committed profile cannot drift."""
import json, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = json.loads(open(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'profiles', 'r04_bench.json')).read().strip().splitlines()[-1])
print('step: %.3f ms, %.1f images/s; traffic %.2f GB = %.2f x algorithmic (%.2f TB/s); fp32 %.1f images/s; cpu %.2f images/s' % (
    d['ms_per_step'], d['value'], d['roofline_step']['traffic'] / 1e9, d['roofline_step']['traffic_over_algorithmic'],
    d['roofline_step']['traffic_gbs'] / 1e3, d['extra']['fp32']['images_per_s'], d['cpu_baseline']['value']))
print('dominant_by_step_cost:', d['dominant_by_step_cost']['family'], d['dominant_by_step_cost']['step_ms_saved_without'], ' by kernel time:', d['dominant_by_kernel_time']['step_kernel_us'])
for key in ['roofline', 'roofline_wgrad', 'roofline_bwd_fused', 'roofline_conv_small', 'roofline_ram', 'roofline_conv64']:
    r = d.get(key)
    if not r:
        continue
    unit = r['unit']
    print('%-20s %-10s launches %2d  in step %7.1f %s = %.3f | alone %7.1f = %.3f | traffic %.1f / %.1f MB = %.2f | mfma_busy %.2f | lds conflicts %.4f | step cost %s' % (
        key, r['family'], r['launches_per_step'], r['achieved'], unit, r['frac'], r['alone']['achieved'], r['alone']['frac'],
        r['traffic'] / 1e6, r['avg_algorithmic_bytes'] / 1e6, r['traffic_over_algorithmic'], r.get('mfma_busy', 0), r.get('lds_conflict_ratio', 0),
        r.get('step_cost_ms', '-')))
