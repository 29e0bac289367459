"""Idle time on each hardware queue inside the training step, from a rocprofv3 kernel trace (CSV) of bench.py: per queue, the kernels of the
LAST complete step in start order, the gaps between one kernel's end and the next one's start, and which launches sit behind the largest
gaps.  usage: lane_gaps.py <kernel_trace.csv> [kernels per step on the main queue, default: detect by the RAM kernel]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
key = lambda r, *names: next(r[n] for n in names if n in r)
ev = []
for r in rows:
    ev.append(dict(q=key(r, 'Queue_Id', 'queue_id'), name=key(r, 'Kernel_Name', 'kernel_name'), t0=int(key(r, 'Start_Timestamp', 'start_timestamp')),
                   t1=int(key(r, 'End_Timestamp', 'end_timestamp'))))
ev.sort(key=lambda e: e['t0'])
# step boundaries: adam_update_kernel ends a step
ends = [e['t1'] for e in ev if 'adam_update_kernel' in e['name']]
assert len(ends) >= 3, len(ends)
a, b = ends[-3], ends[-2]                                   # one complete step in the steady state
step = [e for e in ev if a <= e['t0'] < b]
print('step: %.1f us, %d launches' % ((b - a) / 1e3, len(step)))
byq = collections.defaultdict(list)
for e in step:
    byq[e['q']].append(e)
import re
def short(n):
    n = n.replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '')
    n = re.sub(r'^_ZN12_GLOBAL__N_1\d+', '', n)
    return re.sub(r'\(.*$', '', n)[:60]
for q, es in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    busy = sum(e['t1'] - e['t0'] for e in es)
    gaps = [(es[i + 1]['t0'] - es[i]['t1'], es[i], es[i + 1]) for i in range(len(es) - 1)]
    small = [g for g in gaps if 0 <= g[0] < 20000]
    print('queue %s: %d launches, busy %.0f us, first start %+.0f us, last end %+.0f us; %d gaps < 20 us sum %.0f us (median %.1f us)' % (
        q, len(es), busy / 1e3, (es[0]['t0'] - a) / 1e3, (es[-1]['t1'] - a) / 1e3, len(small), sum(g[0] for g in small) / 1e3,
        sorted(g[0] for g in small)[len(small) // 2] / 1e3 if small else 0))
    for g, e0, e1 in sorted(gaps, key=lambda g: -g[0])[:8]:
        print('    gap %7.1f us after %-50s (%.1f us) before %s' % (g / 1e3, short(e0['name']), (e0['t1'] - e0['t0']) / 1e3, short(e1['name'])))
    if len(es) == max(len(v) for v in byq.values()):        # the main queue: every launch with the gap in front of it
        print('  main queue, launch by launch (start since the step began, gap in front, duration):')
        prev = None
        for e in es:
            print('    %+8.1f us  gap %6.1f  dur %6.1f  %s' % ((e['t0'] - a) / 1e3, ((e['t0'] - prev) / 1e3) if prev else 0.0, (e['t1'] - e['t0']) / 1e3, short(e['name'])))
            prev = e['t1']
for q, es in byq.items():
    print('queue %s, last launches:' % q)
    for e in es[-10:]:
        print('    %+8.1f .. %+8.1f us  %s' % ((e['t0'] - a) / 1e3, (e['t1'] - a) / 1e3, short(e['name'])))
