#!/bin/bash
# round 3, batch 5 (GPU box): A/B of this build against the round-2 engine on one box
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03_batch5; mkdir -p $O
S="PMC_NUM_GROUPS=1,PMC_SERIAL_WALKS=1,PMC_TIMING_DUMP=1"
python tools/sweep.py --packets 1e8 default libpmc_r02.so default libpmc_r02.so default,$S,PMC_WALK_BLOCKS_PER_CU=3 libpmc_r02.so,$S,PMC_WALK_BLOCKS_PER_CU=3 > $O/sweep.txt 2>&1
grep -v "amdgpu.ids" $O/sweep.txt | grep "pkt/s\|peel\|PMC_TIMING" | awk '/pkt/ {print last} !/PMC_TIMING/ {print} {last=$0}'
