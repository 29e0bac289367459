#!/bin/bash
# ON THE GPU BOX: rd_ram_mix alone (wall clock per call, HIP events) under the debug library's switch RD_RAM_DFT: 1 = the row pass of
# uint8 images on the matrix cores (csrc/ram_dft.hip), 0 = the FFT row pass.  Per-kernel GPU times: scripts/ram_prof.sh.
cd $GRAFT_REPO_ROOT
export RAMDSIR_DEBUG_LIB=1
for a in "u8 400" "u8 256" "u8 384" "u8 512" "f32 400"; do
  for w in 1 0; do echo -n "RD_RAM_DFT=$w: "; RD_RAM_DFT=$w python scripts/ram_bench.py $a | tail -1; done
done
