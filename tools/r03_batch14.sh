#!/bin/bash
# round 3, batch 14 (GPU box): launch geometry after the kernel split (the transition-side kernels hold less LDS now)
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03_batch14; mkdir -p $O
python tools/sweep.py --packets 1e8 default default \
  default,PMC_WALK_BLOCKS_PER_CU=2 default,PMC_WALK_BLOCKS_PER_CU=3 default,PMC_PEEL_BLOCKS_PER_CU=2 default,PMC_WALK_BLOCKS_PER_CU=2,PMC_PEEL_BLOCKS_PER_CU=2 \
  default,PMC_NUM_GROUPS=2 default,PMC_NUM_GROUPS=4 default,PMC_NUM_GROUPS=2,PMC_WALK_BLOCKS_PER_CU=2 default,PMC_NUM_GROUPS=4,PMC_NUM_SLOTS=12582912 \
  default,PMC_TRANSITION_BLOCKS_PER_CU=2 default,PMC_TRANSITION_BLOCKS_PER_CU=8 default,PMC_LAUNCH_BLOCKS_PER_CU=2 default,PMC_LAUNCH_BLOCKS_PER_CU=8 \
  default,PMC_CYCLE_BLOCKS_PER_CU=2 default,PMC_CYCLE_BLOCKS_PER_CU=8 default,PMC_CYCLE_BLOCKS_PER_CU=16 \
  default,PMC_NUM_SLOTS=6291456 default,PMC_NUM_SLOTS=12582912 default > $O/sweep.txt 2>&1
grep -v "amdgpu.ids" $O/sweep.txt | grep "pkt/s" | cut -c1-150
