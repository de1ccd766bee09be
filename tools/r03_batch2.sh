#!/bin/bash
# round 3, batch 2 (GPU box): perturbation sweep, second part (gathers that the compiler keeps; dose response of the VALU load)
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03_batch2; mkdir -p $O
S="PMC_NUM_GROUPS=1,PMC_SERIAL_WALKS=1,PMC_TIMING_DUMP=1"
python tools/sweep.py --packets 5e7 \
  default,$S,PMC_WALK_BLOCKS_PER_CU=3 \
  libpmc_pert_gather_1.so,$S,PMC_WALK_BLOCKS_PER_CU=3 \
  libpmc_pert_gather_2.so,$S,PMC_WALK_BLOCKS_PER_CU=3 \
  libpmc_pert_valu_16.so,$S,PMC_WALK_BLOCKS_PER_CU=3 \
  libpmc_pert_valu_96.so,$S,PMC_WALK_BLOCKS_PER_CU=3 \
  libpmc_pert_gather_2.so,$S,PMC_WALK_BLOCKS_PER_CU=1 \
  libpmc_pert_valu_96.so,$S,PMC_WALK_BLOCKS_PER_CU=1 \
  libpmc_pert_valu_96.so,$S,PMC_WALK_BLOCKS_PER_CU=2 \
  default,$S,PMC_WALK_BLOCKS_PER_CU=3,PMC_PEEL_BLOCKS_PER_CU=2 \
  libpmc_pert_gather_2.so,$S,PMC_WALK_BLOCKS_PER_CU=3,PMC_PEEL_BLOCKS_PER_CU=2 \
  default \
  libpmc_pert_gather_1.so \
  libpmc_pert_gather_2.so \
  > $O/sweep.txt 2>&1
grep -v "^PMC_TIMING peel 1[0-9]\.\|amdgpu.ids" $O/sweep.txt | awk 'NR%3!=1 || /pkt/' | tail -60
