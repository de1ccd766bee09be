#!/bin/bash
# round 3, batch 24 (GPU box): propagation kernel with LDS task queues in ONE workgroup of 512 / 768 lanes per CU (8 / 12 waves sharing one
# coordinate table; fewer pass-1 records to make room)
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03_batch24; mkdir -p $O
S="PMC_NUM_GROUPS=1,PMC_SERIAL_WALKS=1,PMC_TIMING_DUMP=1"
python tools/sweep.py --packets 1e8 default,$S,PMC_WALK_BLOCKS_PER_CU=3 libpmc_q768t2.so,$S libpmc_q768t2.so,$S,PMC_PROP_V1=1 libpmc_q768t2r24.so,$S libpmc_q768t2r8.so,$S libpmc_q512t3.so,$S \
   default libpmc_q768t2.so libpmc_q768t2r24.so libpmc_q512t3.so default libpmc_q768t2.so > $O/sweep.txt 2>&1
grep -v "amdgpu.ids" $O/sweep.txt | grep "pkt/s\|PMC_TIMING\|prop" | awk '/pkt/ {print last} !/PMC_TIMING/ {print} {last=$0}' | cut -c1-250
