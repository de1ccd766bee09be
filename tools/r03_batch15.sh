#!/bin/bash
# round 3, batch 15 (GPU box): propagation kernel with per-wave task queues in LDS: parity, then against the round form
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03_batch15; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -m gpu -x -q > $O/gputests.txt 2>&1; echo "gpu tests rc=$?"; tail -5 $O/gputests.txt | cut -c1-300
S="PMC_NUM_GROUPS=1,PMC_SERIAL_WALKS=1,PMC_TIMING_DUMP=1"
python tools/sweep.py --packets 1e8 default,PMC_PROP_V1=1 default libpmc_q8.so libpmc_q16.so libpmc_q24.so default,PMC_PROP_V1=1 default \
   default,PMC_PROP_V1=1,$S default,$S libpmc_q8.so,$S libpmc_q24.so,$S default,PMC_PROP_V1=1,$S,PMC_WALK_BLOCKS_PER_CU=3 > $O/sweep.txt 2>&1
grep -v "amdgpu.ids" $O/sweep.txt | grep "pkt/s\|PMC_TIMING\|prop" | awk '/pkt/ {print last} !/PMC_TIMING/ {print} {last=$0}' | cut -c1-250
