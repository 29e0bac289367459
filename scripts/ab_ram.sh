#!/bin/bash
# ON THE GPU BOX: rd_ram_mix alone under the debug library's switches.  RD_RAM_WAVE bits 0 / 1 / 2 = the wave-level kernel for the row
# forward / column mix / row inverse pass (csrc/ram_wavefft.h); 0 = the Stockham kernels for all three.
cd $GRAFT_REPO_ROOT
export RAMDSIR_DEBUG_LIB=1
for spec in "wave-all:RD_RAM_WAVE=7" "stockham:RD_RAM_WAVE=0" "wave-A:RD_RAM_WAVE=1" "wave-B:RD_RAM_WAVE=2" "wave-C:RD_RAM_WAVE=4"; do
  name=${spec%%:*}; vars=${spec#*:}
  ( IFS=','; for kv in $vars; do export "$kv"; done; echo -n "$name: "; python scripts/ram_bench.py u8 400 2>&1 | tail -1 )
done
for a in "f32 400" "u8 256" "u8 384" "u8 512"; do
  for w in 7 0; do echo -n "RD_RAM_WAVE=$w: "; RD_RAM_WAVE=$w python scripts/ram_bench.py $a | tail -1; done
done
