"""MFMA utilisation per kernel family of bench.py from the SQ counters of one eager step:
    rocprofv3 -i scripts/pmc_sq.txt --kernel-trace --output-format csv -d <dir> -o p -- python3 scripts/pmc_step.py bf16 400 2
    python3 scripts/mfma_busy.py <dir> [tag]      ->  profiles/<tag>_mfma_busy.json
mfma_busy = sum SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8 (the counter is
summed over the 8 XCDs; MI355X_MICROARCH.md: SQ_VALU_MFMA_BUSY_CYCLES counts cycles, 32 per v_mfma_f32_32x32x16_bf16) -- the
fraction of the CHIP's matrix-pipe time the family's kernels keep busy while they run (single stream, nothing beside them);
`mfma_busy_occupied` divides by the SIMDs of the CUs the launch can occupy (min(workgroups, 256) CUs) instead."""
import collections, csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as Bn
d = sys.argv[1]
tag = sys.argv[2] if len(sys.argv) > 2 else 'r04'
# the counter passes are separate runs of the same launch list (dispatch ids can shift between them): aggregate per
# (kernel, grid, workgroup size) and counter, then combine the per-dispatch MEANS
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        acc[(r['Kernel_Name'], r['Grid_Size'], r['Workgroup_Size'])][r['Counter_Name']].append(float(r['Counter_Value']))
DEM = {'conv_pf_kernelIDF16bLi9ELi2E': 'conv_pf_kernel<__bf16, 9, 2', '11conv_kernelIDF16bLi9ELi2E': 'conv_kernel<__bf16, 9, 2'}
out = {'collected': 'rocprofv3 -i scripts/pmc_sq.txt on scripts/pmc_step.py bf16 400 2 (one stream), scripts/mfma_busy.py'}
mean = lambda v: sum(v) / len(v)
for fam, spec in Bn.FAMILIES.items():
    busy = cyc = occ = n = 0.0
    conf = ldsact = 0.0
    for (name, grid, wgs), c in acc.items():
        if not any(DEM.get(s, s) in name for s in spec['symbols']):
            continue
        if 'SQ_VALU_MFMA_BUSY_CYCLES' not in c or 'GRBM_GUI_ACTIVE' not in c:
            continue
        k = len(c['GRBM_GUI_ACTIVE'])
        kc = mean(c['GRBM_GUI_ACTIVE']) / 8.0
        cus = min(int(grid) // max(int(wgs), 1), 256)
        busy += k * mean(c['SQ_VALU_MFMA_BUSY_CYCLES'])
        cyc += k * 1024.0 * kc
        occ += k * 4.0 * cus * kc
        n += k
        if 'SQ_LDS_BANK_CONFLICT' in c and 'SQ_LDS_IDX_ACTIVE' in c:
            conf += k * mean(c['SQ_LDS_BANK_CONFLICT'])
            ldsact += k * mean(c['SQ_LDS_IDX_ACTIVE'])
    if n:
        out[fam] = dict(mfma_busy=round(busy / cyc, 4), mfma_busy_occupied=round(busy / occ, 4), dispatches=int(n))
        if ldsact:
            out[fam]['lds_conflict_ratio'] = round(conf / ldsact, 4)          # SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
json.dump(out, open(os.path.join(ROOT, 'profiles', tag + '_mfma_busy.json'), 'w'), indent=1)
print(json.dumps(out, indent=1))
