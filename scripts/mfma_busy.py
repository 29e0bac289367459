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
# (kernel, grid, workgroup size) and counter, then combine the per-dispatch MEANS (bench.sq_aggregate, shared with the live collection)
out = {'collected': 'rocprofv3 -i scripts/pmc_sq.txt on scripts/pmc_step.py bf16 400 2 (one stream), scripts/mfma_busy.py'}
out.update(Bn.sq_aggregate(Bn.read_counter_dir(d)))
json.dump(out, open(os.path.join(ROOT, 'profiles', tag + '_mfma_busy.json'), 'w'), indent=1)
print(json.dumps(out, indent=1))
