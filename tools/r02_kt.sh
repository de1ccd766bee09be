#!/bin/bash
# kernel trace of one step of 5e7 packets, one slot group, walk kernels in series (per-kernel times without overlap)
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/${RUN_NAME:-r02_kt}; mkdir -p $O
(cd /tmp && PMC_NUM_GROUPS=1 PMC_SERIAL_WALKS=1 PMC_WALK_BLOCKS_PER_CU=3 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python $R/bench.py --steps 1 --warmup 0 --packets 5e7 --no-cpu-baseline --no-secondary > $O/kt.log 2>&1)
find $O/kt -name "*kernel_stats.csv" -exec cat {} \; | head -6 | cut -c1-160
find $O -name "*kernel_trace.csv" -size +20M -delete
