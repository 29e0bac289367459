"""Summarise rocprofv3 --pmc output of scripts/one_layer.py runs: counters of the repeated launch only."""
import csv, glob, collections, os, sys
for d in sorted(glob.glob(sys.argv[1])):
    if not os.path.isdir(d): continue
    tr = list(csv.DictReader(open(d + '/pmc_1/p_kernel_trace.csv')))
    want = sys.argv[2] if len(sys.argv) > 2 else ''
    last = [r for r in tr if want in r['Kernel_Name']][-1]
    kname = last['Kernel_Name']; grid = (last['Grid_Size_X'], last['Grid_Size_Y'], last['Grid_Size_Z'])
    durs = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in tr if r['Kernel_Name'] == kname and (r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z']) == grid]
    m = collections.defaultdict(list)
    for f in glob.glob(d + '/pmc_*/p_counter_collection.csv'):
        trf = list(csv.DictReader(open(f.replace('counter_collection', 'kernel_trace'))))
        idsf = set(r['Dispatch_Id'] for r in trf if r['Kernel_Name'] == kname and (r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z']) == grid)
        for r in csv.DictReader(open(f)):
            if r['Dispatch_Id'] in idsf: m[r['Counter_Name']].append(float(r['Counter_Value']))
    med = {k: sorted(v)[len(v) // 2] for k, v in m.items()}
    g = lambda k: med.get(k, 0)
    print('==', os.path.basename(d), kname[:70], 'grid', grid, 'n', len(durs), 'median us %.1f' % sorted(durs)[len(durs) // 2])
    w = max(g('SQ_WAVES'), 1); wc = max(g('SQ_WAVE_CYCLES'), 1)
    print('   waves %d  WAIT_ANY %.0f%% WAIT_INST %.0f%% ACTIVE %.0f%% | per wave: VALU %.0f SALU %.0f LDS %.0f VMEM_RD %.0f VMEM_WR %.0f MFMA %.0f | wave_cycles/wave %.0f' % (
        w, 100 * g('SQ_WAIT_ANY') / wc, 100 * g('SQ_WAIT_INST_ANY') / wc, 100 * g('SQ_ACTIVE_INST_ANY') / wc, g('SQ_INSTS_VALU') / w, g('SQ_INSTS_SALU') / w,
        g('SQ_INSTS_LDS') / w, g('SQ_INSTS_VMEM_RD') / w, g('SQ_INSTS_VMEM_WR') / w, g('SQ_INSTS_MFMA') / w, wc / w))
    print('   ACTIVE_VALU %.3g ACTIVE_LDS %.3g LDS_BANK_CONFLICT %.3g LDS_IDX_ACTIVE %.3g FETCH %.0f KB WRITE %.0f KB TCC hit %.3g miss %.3g GUI %.3g BUSY %.3g' % (
        g('SQ_ACTIVE_INST_VALU'), g('SQ_ACTIVE_INST_LDS'), g('SQ_LDS_BANK_CONFLICT'), g('SQ_LDS_IDX_ACTIVE'), g('FETCH_SIZE'), g('WRITE_SIZE'), g('TCC_HIT_sum'), g('TCC_MISS_sum'),
        g('GRBM_GUI_ACTIVE'), g('SQ_BUSY_CYCLES')))
