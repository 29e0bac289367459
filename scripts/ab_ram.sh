#!/bin/bash
# ON THE GPU BOX: rd_ram_mix alone under the debug library's geometry switches.
cd $GRAFT_REPO_ROOT
export RAMDSIR_DEBUG_LIB=1
for spec in "default:X=1" "fwd1:RD_RAM_ROWS_FWD=1" "kt2:RD_RAM_KT=2" "kt1:RD_RAM_KT=1" "inv1:RD_RAM_ROWS_INV=1" "inv4:RD_RAM_ROWS_INV=4" "fwd1kt2:RD_RAM_ROWS_FWD=1,RD_RAM_KT=2" "fwd1kt2inv4:RD_RAM_ROWS_FWD=1,RD_RAM_KT=2,RD_RAM_ROWS_INV=4"; do
  name=${spec%%:*}; vars=${spec#*:}
  ( IFS=','; for kv in $vars; do export "$kv"; done; echo -n "$name: "; python scripts/ram_bench.py u8 400 2>&1 | tail -1 )
done
python scripts/ram_bench.py f32 400 | tail -1
python scripts/ram_bench.py u8 256 | tail -1
python scripts/ram_bench.py u8 512 | tail -1
