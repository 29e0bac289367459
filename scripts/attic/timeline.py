"""Per-stream timeline of the LAST full step in a rocprofv3 --kernel-trace CSV: busy time per queue, concurrency histogram,
gaps on the main queue, the tail.  usage: timeline.py <kernel_trace.csv> [n_launches_per_step]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# step boundaries: the Adam kernel ends a step
adam = [i for i, r in enumerate(rows) if 'adam' in r['Kernel_Name']]
if len(adam) < 3:
    print('not enough steps'); sys.exit(0)
lo, hi = adam[-3] + 1, adam[-2] + 1          # a step in the middle of the timed region (pack_weights follows adam: include it)
while hi < len(rows) and 'pack_weights' in rows[hi]['Kernel_Name']:
    hi += 1
while 'pack_weights' in rows[lo]['Kernel_Name']:
    lo += 1
step = rows[lo:hi]
t0 = min(int(r['Start_Timestamp']) for r in step); t1 = max(int(r['End_Timestamp']) for r in step)
print('step: %d launches, %.1f us wall' % (len(step), (t1 - t0) / 1e3))
byq = collections.defaultdict(list)
for r in step:
    byq[r['Queue_Id']].append((int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0, r['Kernel_Name']))
for q, ks in sorted(byq.items(), key=lambda kv: -sum(e - s for s, e, _ in kv[1])):
    busy = sum(e - s for s, e, _ in ks)
    print('queue %s: %3d kernels, busy %.0f us, first start %.0f us, last end %.0f us' % (q, len(ks), busy / 1e3, ks[0][0] / 1e3, max(e for _, e, _ in ks) / 1e3))
# concurrency histogram
ev = []
for r in step:
    ev.append((int(r['Start_Timestamp']) - t0, 1)); ev.append((int(r['End_Timestamp']) - t0, -1))
ev.sort()
hist = collections.Counter(); cur = 0; last = 0
for t, d in ev:
    hist[cur] += t - last; last = t; cur += d
print('time with k kernels in flight: ' + '  '.join('k=%d: %.0f us' % (k, v / 1e3) for k, v in sorted(hist.items())))
# 10 coarse slices: which queues are busy
n = 12
print('slice  ' + '  '.join('q%s' % q for q in sorted(byq)))
for i in range(n):
    a, b = (t1 - t0) * i // n, (t1 - t0) * (i + 1) // n
    cells = []
    for q in sorted(byq):
        ov = sum(max(0, min(e, b) - max(s, a)) for s, e, _ in byq[q])
        cells.append('%3.0f%%' % (100.0 * ov / (b - a)))
    print('%5.0f  ' % (a / 1e3) + '  '.join(cells))
# longest kernels under contention
print('longest launches in this step:')
for r in sorted(step, key=lambda r: int(r['Start_Timestamp']) - int(r['End_Timestamp']))[:12]:
    print('  %7.1f us  q%s  @%6.0f  %s' % ((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r['Queue_Id'], (int(r['Start_Timestamp']) - t0) / 1e3, r['Kernel_Name'][:100]))
