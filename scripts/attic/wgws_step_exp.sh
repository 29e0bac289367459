export RAMDSIR_DEBUG_LIB=1
for i in 1 2 3; do
for x in 0 4; do
  ms=$(RD_WGWS_EXP=$x python3 bench.py --no-cpu-baseline --no-fp32-leg --no-ablation --steps 100 --warmup 10 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.readlines()[-1])['ms_per_step'])")
  echo "RD_WGWS_EXP=$x $ms ms/step"
done
done
for x in 0 4; do
  for sc in 80 64; do
  ms=$(RD_WGWS_EXP=$x RD_SIDE_CUS=$sc python3 bench.py --no-cpu-baseline --no-fp32-leg --no-ablation --steps 100 --warmup 10 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.readlines()[-1])['ms_per_step'])")
  echo "RD_WGWS_EXP=$x side_cus=$sc $ms ms/step"
  done
done
