#!/bin/bash
# round 2, experiment 1: device cell numbering (sibling groups scattered over the table) against L2-channel hot spots
export TMPDIR=/tmp
O=gpurun_out/r02_exp1; rm -rf $O; mkdir -p $O
timeout 900 python tools/sweep.py default default,PMC_CELL_SHUFFLE=2 default,PMC_CELL_SHUFFLE=3 default,PMC_CELL_SHUFFLE=4 \
   default,PMC_CELL_SHUFFLE=6 default,PMC_CELL_SHUFFLE=8 default,PMC_CELL_SHUFFLE=0 \
   default,PMC_NUM_GROUPS=1 default,PMC_NUM_GROUPS=1,PMC_CELL_SHUFFLE=2 default,PMC_NUM_GROUPS=1,PMC_CELL_SHUFFLE=3 \
   libpmc_w4.so,PMC_WALK_BLOCKS_PER_CU=4,PMC_NUM_GROUPS=1 libpmc_w4.so,PMC_WALK_BLOCKS_PER_CU=4,PMC_NUM_GROUPS=1,PMC_CELL_SHUFFLE=2 \
   libpmc_prof.so,PMC_NUM_GROUPS=1,PMC_PROFILE_DUMP=1 libpmc_prof.so,PMC_NUM_GROUPS=1,PMC_PROFILE_DUMP=1,PMC_CELL_SHUFFLE=2 \
   > $O/sweep.txt 2> $O/sweep.err
cat $O/sweep.txt; grep PMC_PROFILE $O/sweep.err
PMC_CELL_SHUFFLE=2 timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_shuffle2.log 2>&1; tail -3 $O/pytest_shuffle2.log
