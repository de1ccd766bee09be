#!/bin/bash
# round 3, batch 10 (GPU box): how many checkpoints pay (HBM stores against re-walked cells, which hit in L2)
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03_batch10; mkdir -p $O
S="PMC_NUM_GROUPS=1,PMC_SERIAL_WALKS=1,PMC_TIMING_DUMP=1,PMC_WALK_BLOCKS_PER_CU=3"
python tools/sweep.py --packets 1e8 libpmc_r02.so,$S default,$S libpmc_ckf24.so,$S libpmc_ck1.so,$S libpmc_ck1t6.so,$S libpmc_ck1t8.so,$S libpmc_ck2f32.so,$S libpmc_r02.so,$S \
    libpmc_r02.so default libpmc_ck1.so libpmc_ck1t8.so libpmc_ck2f32.so libpmc_r02.so default libpmc_ck1.so libpmc_ck1t8.so > $O/sweep.txt 2>&1
grep -v "amdgpu.ids" $O/sweep.txt | grep "pkt/s\|PMC_TIMING\|prop" | awk '/pkt/ {print last} !/PMC_TIMING/ {print} {last=$0}' | cut -c1-250 | grep -v "^    peel.*0\.[23][0-9][0-9]e11 lane-steps/s, lanes 5"
