"""Compare the per-layer rows of scripts/ab_layers_lib.sh's log: mean per (build, layer)."""
import re, collections, sys
cur = None
d = collections.defaultdict(lambda: collections.defaultdict(list))
for l in open(sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/ab_layers.log'):
    m = re.match(r'== (\S+)', l)
    if m:
        cur = m.group(1)
        continue
    m = re.match(r'\s*([0-9.]+) us\s+#\s*(\d+)\s+(\S+)\s+(\S+)\s+(\S+)\s+(\S+)', l)
    if m and cur:
        d[m.group(2) + ' ' + m.group(6)][cur].append(float(m.group(1)))
builds = sorted({b for v in d.values() for b in v})
tot = {b: 0.0 for b in builds}
for layer, v in sorted(d.items(), key=lambda kv: int(kv[0].split()[0])):
    means = [sum(v[b]) / max(len(v[b]), 1) for b in builds]
    for b, m in zip(builds, means):
        tot[b] += m
    print('%-28s ' % layer + '  '.join('%s %6.1f' % (b, m) for b, m in zip(builds, means)) + '  %+5.1f%%' % (100 * (means[0] - means[1]) / means[1] if len(means) > 1 else 0))
print(tot)
