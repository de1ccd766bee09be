#!/bin/bash
# round 3, batch 7 (GPU box): kernel trace, one slot group with the walk kernels in series (no overlap: durations add up)
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03_batch7; mkdir -p $O
(cd /tmp && PMC_NUM_GROUPS=1 PMC_SERIAL_WALKS=1 PMC_WALK_BLOCKS_PER_CU=3 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python $R/bench.py --steps 1 --warmup 1 --packets 5e7 --no-cpu-baseline --no-secondary > $O/kt.log 2>&1)
find $O -name "*kernel_stats.csv" | head; f=$(find $O -name "*kernel_stats.csv" | head -1); cut -c1-200 $f | head -20
find $O -name "*kernel_trace.csv" -size +20M -delete
