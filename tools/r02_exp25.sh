#!/bin/bash
# round 2, experiment 25: size of the slot pool at the bench's packet count (fewer, larger generations against their tails)
export TMPDIR=/tmp
O=gpurun_out/r02_exp25; rm -rf $O; mkdir -p $O
timeout 1200 python tools/sweep.py --packets 1e8 default default,PMC_NUM_SLOTS=16777216 default,PMC_NUM_SLOTS=33554432 default,PMC_NUM_SLOTS=16777216,PMC_NUM_GROUPS=4 default,PMC_NUM_SLOTS=33554432,PMC_NUM_GROUPS=4 default,PMC_NUM_SLOTS=4194304 > $O/sweep.txt 2> $O/sweep.err
cut -c1-170 $O/sweep.txt
