#!/bin/bash
# round 3, batch 27 (GPU box): with the 768-lane propagation workgroup: peel-off workgroup size, steps between round checks, peel round threshold,
# workgroups of the transition-side kernels
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03_batch27; mkdir -p $O
python tools/sweep.py --packets 1e8 default default libpmc_p384.so libpmc_p640.so libpmc_ws8.so libpmc_ws2.so libpmc_pr4.so libpmc_pr16.so \
   default,PMC_TRANSITION_BLOCKS_PER_CU=2 default,PMC_TRANSITION_BLOCKS_PER_CU=8 default,PMC_CYCLE_BLOCKS_PER_CU=2 default,PMC_CYCLE_BLOCKS_PER_CU=8 default,PMC_LAUNCH_BLOCKS_PER_CU=2 default,PMC_LAUNCH_BLOCKS_PER_CU=8 default > $O/sweep.txt 2>&1
grep -v "amdgpu.ids" $O/sweep.txt | grep "pkt/s" | cut -c1-160
