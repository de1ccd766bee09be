#!/bin/bash
# round 3, batch 8 (GPU box): section timers of the transition and launch kernels (profiling build)
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03_batch8; mkdir -p $O
python tools/sweep.py --packets 5e7 libpmc_prof.so,PMC_NUM_GROUPS=1,PMC_SERIAL_WALKS=1,PMC_TIMING_DUMP=1,PMC_PROFILE_DUMP=1,PMC_WALK_BLOCKS_PER_CU=3 > $O/prof.txt 2>&1
grep -v "amdgpu.ids" $O/prof.txt | grep "PMC_PROFILE\|pkt/s\|PMC_TIMING" | cut -c1-400
