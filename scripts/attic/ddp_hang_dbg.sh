#!/bin/bash
# stress: the two-rank ddp worker pair N times; SIGABRT after 90 s makes faulthandler dump every thread's Python stack of a hung rank
cd $GRAFT_REPO_ROOT
N=${1:-10}
for i in $(seq 1 $N); do
  for r in 0 1; do
    timeout -s ABRT 90 python -X faulthandler tests/ddp_worker.py --rank $r --world 2 --port $((29611 + i)) --graph 0 > gpurun_out/ddp_dbg_${i}_$r.log 2>&1 &
  done
  wait
  ok=$(grep -l DDPRESULT gpurun_out/ddp_dbg_${i}_0.log gpurun_out/ddp_dbg_${i}_1.log 2>/dev/null | wc -l)
  echo "iter $i: $ok of 2 ranks finished"
  if [ "$ok" = "2" ]; then rm -f gpurun_out/ddp_dbg_${i}_*.log; fi
done
