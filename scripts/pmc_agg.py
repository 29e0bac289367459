"""Aggregate rocprofv3 counter_collection CSVs per (kernel, grid): mean counter value per dispatch.
usage: pmc_agg.py <dir> [name-filter]"""
import csv, glob, os, sys, collections
d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ''
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if flt and flt not in k:
            continue
        key = (k[:90], r.get('Grid_Size', ''), r.get('LDS_Block_Size', ''))
        acc[key][r['Counter_Name']].append(float(r['Counter_Value']))
rows = []
for key, cs in acc.items():
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    rows.append((m.get('GRBM_GUI_ACTIVE', 0) * len(next(iter(cs.values()))), key, m, len(next(iter(cs.values())))))
rows.sort(key=lambda r: -r[0])
for _, key, m, n in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    print('%s grid=%s lds=%s n=%d' % (key[0], key[1], key[2], n))
    wc = m.get('SQ_WAVE_CYCLES', 0) or 1
    print('   ' + '  '.join('%s=%.3g' % (c.replace('SQ_', ''), v) for c, v in sorted(m.items())))
    if 'SQ_WAIT_ANY' in m:
        # MFMA utilisation: SQ_VALU_MFMA_BUSY_CYCLES (cycles, summed over the SIMDs) / (1024 SIMDs x kernel cycles), kernel cycles =
        # GRBM_GUI_ACTIVE / 8 (summed over the 8 XCDs).  (Round 2 printed mfma_busy / (4 * SQ_BUSY_CYCLES), which exceeds 1:
        # SQ_BUSY_CYCLES is not a per-SIMD time base.)
        kc = max(m.get('GRBM_GUI_ACTIVE', 0) / 8.0, 1.0)
        print('   frac of wave cycles: wait_any %.2f  wait_inst_any %.2f  active_inst_any %.2f  wait_inst_lds %.2f ; mfma_busy (chip) %.3f'
              % (m['SQ_WAIT_ANY'] / wc, m['SQ_WAIT_INST_ANY'] / wc, m['SQ_ACTIVE_INST_ANY'] / wc, m.get('SQ_WAIT_INST_LDS', 0) / wc,
                 m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (1024.0 * kc)))
