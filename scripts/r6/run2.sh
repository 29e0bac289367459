#!/bin/bash
# round 6, GPU call 2: the swap probe's exact-sequence mode; which tensor moves between the old-swap and the new build; the new
# weight-gradient kernel: op tests, per-layer timing against the old kernels (debug library, RD_WG_SYM=0)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r6b; mkdir -p $O
P=scripts/probe/swap_probe.bin
( echo "== alone"; timeout 200 $P 20 20000
  echo "== three processes at once"
  for k in 1 2 3; do (timeout 300 $P 20 20000 > $O/swap_$k.txt 2>&1) & done; wait
  cat $O/swap_1.txt $O/swap_2.txt $O/swap_3.txt ) > $O/swap_probe.txt 2>&1
L=ram-dsir_amd/ramdsir/libramdsir_hip.so
cp $L /tmp/keep.so
timeout 300 python3 scripts/r6/ab_bits.py dump $O/bits_new.json 4 > $O/ab_bits.txt 2>&1
cp ab/swap_old.so $L
timeout 300 python3 scripts/r6/ab_bits.py dump $O/bits_old.json 4 >> $O/ab_bits.txt 2>&1
cp /tmp/keep.so $L
python3 scripts/r6/ab_bits.py cmp $O/bits_old.json $O/bits_new.json >> $O/ab_bits.txt 2>&1
timeout 900 python3 -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "wgrad" > $O/pytest_wgrad.txt 2>&1
tail -5 $O/pytest_wgrad.txt
timeout 300 python3 scripts/layer_bench.py bf16 400 200 > $O/layer_new.txt 2>&1
RAMDSIR_DEBUG_LIB=1 RD_WG_SYM=0 timeout 300 python3 scripts/layer_bench.py bf16 400 200 > $O/layer_old_dbg.txt 2>&1
RAMDSIR_DEBUG_LIB=1 timeout 300 python3 scripts/layer_bench.py bf16 400 200 > $O/layer_new_dbg.txt 2>&1
grep "rd_wgrad" $O/layer_new.txt | head -30
echo ---- old
grep "rd_wgrad" $O/layer_old_dbg.txt | head -30
cat $O/ab_bits.txt
tail -12 $O/swap_probe.txt
