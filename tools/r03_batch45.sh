#!/bin/bash
# round 3, batch 45 (GPU box): sparse generations -- walks per lane that size the walk kernels' grids
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03_batch45; mkdir -p $O
python tools/sweep.py --packets 1e8 default default default,PMC_LIST_TASKS_PER_LANE=2 default,PMC_LIST_TASKS_PER_LANE=3 default,PMC_LIST_TASKS_PER_LANE=4 default default,PMC_LIST_TASKS_PER_LANE=2 default,PMC_LIST_TASKS_PER_LANE=3 default,PMC_LIST_TASKS_PER_LANE=4 > $O/sweep.txt 2>&1; grep "pkt/s" $O/sweep.txt
python tools/sweep.py --packets 1e7 default default default,PMC_LIST_TASKS_PER_LANE=2 default,PMC_LIST_TASKS_PER_LANE=3 default,PMC_LIST_TASKS_PER_LANE=4 default default,PMC_LIST_TASKS_PER_LANE=2 default,PMC_LIST_TASKS_PER_LANE=3 default,PMC_LIST_TASKS_PER_LANE=4 > $O/sweep1e7.txt 2>&1; grep "pkt/s" $O/sweep1e7.txt
