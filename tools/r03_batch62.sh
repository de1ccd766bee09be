#!/bin/bash
# round 3, batch 62 (GPU box): propagation kernel as ONE 768-lane workgroup per CU WITH per-wave task queues in LDS (32 records; two or one pass-1 segments to make room)
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03_batch62; mkdir -p $O
PMC_LIBRARY=$R/skirt9_amd/lib/libpmc_q2.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py -m gpu -x -q > $O/pytest.log 2>&1; grep -E "passed|failed|Error" $O/pytest.log | tail -3
python tools/sweep.py --packets 1e8 default default libpmc_q2.so libpmc_q2.so,PMC_PROP_NO_QUEUE=1 libpmc_q1.so libpmc_q2r8.so libpmc_q2r24.so default libpmc_q2.so > $O/sweep.txt 2>&1; grep -A1 "pkt/s" $O/sweep.txt
python tools/sweep.py --packets 5e7 default,PMC_NUM_GROUPS=1,PMC_SERIAL_WALKS=1,PMC_TIMING_DUMP=1 libpmc_q2.so,PMC_NUM_GROUPS=1,PMC_SERIAL_WALKS=1,PMC_TIMING_DUMP=1 > $O/serial.txt 2>&1; grep -A1 "pkt/s" $O/serial.txt
