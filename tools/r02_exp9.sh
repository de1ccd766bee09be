#!/bin/bash
# round 2, experiment 9: ablations of the split walk kernels (one slot group, peel-off and propagation kernels in series)
export TMPDIR=/tmp
O=gpurun_out/r02_exp9; rm -rf $O; mkdir -p $O
E=PMC_NUM_GROUPS=1,PMC_SERIAL_WALKS=1,PMC_TIMING_DUMP=1
timeout 900 python tools/sweep.py default,$E libpmc_noslow.so,$E libpmc_nodesc.so,$E libpmc_s8.so,$E libpmc_r8.so,$E libpmc_r32.so,$E \
  default,$E,PMC_PEEL_BLOCKS_PER_CU=1 default,$E,PMC_PEEL_BLOCKS_PER_CU=2 default,$E,PMC_WALK_BLOCKS_PER_CU=1 default,$E,PMC_WALK_BLOCKS_PER_CU=2 \
  > $O/sweep.txt 2> $O/sweep.err
paste -d' ' <(cut -c1-30,100-190 $O/sweep.txt) <(grep PMC_TIMING $O/sweep.err | awk 'NR%3==0')
