#!/bin/bash
# round 3, batch 2 (GPU box): perturbation sweep, second part: extra gathers that HIT in cache (first 128 KB of the table),
# gathers near the cell, dose response of the VALU load, and the reduced scene (L2-resident table)
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03_batch2; mkdir -p $O
S="PMC_NUM_GROUPS=1,PMC_SERIAL_WALKS=1,PMC_TIMING_DUMP=1"
python tools/sweep.py --packets 5e7 \
  default,$S,PMC_WALK_BLOCKS_PER_CU=3 \
  libpmc_pert_gather_3.so,$S,PMC_WALK_BLOCKS_PER_CU=3 \
  libpmc_pert_gather_4.so,$S,PMC_WALK_BLOCKS_PER_CU=3 \
  libpmc_pert_valu_16.so,$S,PMC_WALK_BLOCKS_PER_CU=3 \
  libpmc_pert_valu_96.so,$S,PMC_WALK_BLOCKS_PER_CU=3 \
  libpmc_pert_gather_3.so,$S,PMC_WALK_BLOCKS_PER_CU=1 \
  libpmc_pert_valu_96.so,$S,PMC_WALK_BLOCKS_PER_CU=1 \
  libpmc_pert_gather_3.so \
  libpmc_pert_valu_96.so \
  > $O/sweep.txt 2>&1
python tools/sweep.py --packets 5e7 --ski tests/ski/cfg2small.ski \
  default,$S,PMC_WALK_BLOCKS_PER_CU=3 \
  libpmc_pert_gather_2.so,$S,PMC_WALK_BLOCKS_PER_CU=3 \
  libpmc_pert_valu_96.so,$S,PMC_WALK_BLOCKS_PER_CU=3 \
  >> $O/sweep.txt 2>&1
grep -v "amdgpu.ids" $O/sweep.txt | grep "pkt/s\|PMC_TIMING" | awk '/pkt/ {print last; print $0} {last=$0}'
