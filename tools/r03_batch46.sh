#!/bin/bash
# round 3, batch 46 (GPU box): the new list-form test; launch / transition kernels capped at 168 registers (three waves per SIMD, some scratch)
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03_batch46; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_multi.py -m gpu -x -q > $O/pytest.log 2>&1; grep -E "passed|failed|Error|error" $O/pytest.log | tail -5
python tools/sweep.py --packets 1e8 default default libpmc_l3.so libpmc_t3.so default libpmc_l3.so libpmc_t3.so > $O/sweep.txt 2>&1; grep "pkt/s" $O/sweep.txt
