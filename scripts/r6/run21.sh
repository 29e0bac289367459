#!/bin/bash
# fragment read order in conv_ws: parity, per-layer and step A/B
mkdir -p gpurun_out/r6n
timeout 900 python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_bitwise_golden.py -m gpu -x -q -k "conv or golden or bitwise" 2>&1 | tail -3 > gpurun_out/r6n/pytest.txt
cat gpurun_out/r6n/pytest.txt
ROWS=60 bash scripts/r6/ab_layers.sh "conv_kernel<bf16,9" ab/off32.so ab/rdord.so > gpurun_out/r6n/layers.txt 2>&1
bash scripts/r6/ab_many.sh 3 ab/off32.so ab/rdord.so 2>&1 | tee gpurun_out/r6n/step.txt
export RAMDSIR_DEBUG_LIB=1 RD_CONV_WS_TRACE_MIN=1
timeout 300 python3 scripts/r6/ws_trace2.py fwd dec.convu2.conv3 10 > gpurun_out/r6n/fine_fwd.txt 2>&1
timeout 300 python3 scripts/r6/ws_trace2.py dgrad dec.convu2.conv3 10 > gpurun_out/r6n/fine_dgrad.txt 2>&1
