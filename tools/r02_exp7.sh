#!/bin/bash
# round 2, experiment 7: event census of the walk kernels (literal-algorithm lanes, descents); kernel trace without overlap
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r02_exp7; rm -rf $O; mkdir -p $O
timeout 600 python tools/sweep.py libpmc_census.so,PMC_NUM_GROUPS=1,PMC_PROFILE_DUMP=1 default,PMC_NUM_GROUPS=1 > $O/sweep.txt 2> $O/sweep.err
cat $O/sweep.txt; grep PMC_PROFILE $O/sweep.err
(cd /tmp && PMC_NUM_GROUPS=1 PMC_SERIAL_WALKS=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python $R/bench.py --steps 1 --warmup 0 --packets 2e7 --no-cpu-baseline > $O/kt.log 2>&1)
find $O/kt -name "*kernel_stats.csv" -exec cat {} \; | head -8
find $O -name "*kernel_trace.csv" -size +20M -delete
