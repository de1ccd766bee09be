#!/bin/bash
# round 3, batch 29 (GPU box): how exclusive should the kernels run?  (the 768-lane propagation workgroup fills a CU's LDS on its own)
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03_batch29; mkdir -p $O
python tools/sweep.py --packets 1e8 default default default,PMC_SERIAL_WALKS=1 default,PMC_SERIAL_WALKS=1,PMC_NUM_GROUPS=2 default,PMC_SERIAL_WALKS=1,PMC_NUM_GROUPS=4 default,PMC_NUM_GROUPS=2 \
   default,PMC_PEEL_BLOCKS_PER_CU=2 default,PMC_TRANSITION_BLOCKS_PER_CU=2,PMC_LAUNCH_BLOCKS_PER_CU=2,PMC_CYCLE_BLOCKS_PER_CU=2 default,PMC_TRANSITION_BLOCKS_PER_CU=1,PMC_LAUNCH_BLOCKS_PER_CU=1,PMC_CYCLE_BLOCKS_PER_CU=1 default > $O/sweep.txt 2>&1
grep -v "amdgpu.ids" $O/sweep.txt | grep "pkt/s" | cut -c1-180
