#!/bin/bash
# round 2, experiment 3: where does the time of the split walk kernels go (kernel trace + hardware counters)
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r02_exp3; rm -rf $O; mkdir -p $O
export PMC_NUM_GROUPS=1
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python $R/bench.py --steps 1 --warmup 0 --packets 2e7 --no-cpu-baseline > $O/kt.log 2>&1)
find $O/kt -name "*kernel_stats.csv" -exec cat {} \; | head -20
find $O -name "*kernel_trace.csv" -size +20M -delete
bash tools/pmc_passes.sh --steps 1 --warmup 0 --packets 2e7 --no-cpu-baseline > $O/pmc.log 2>&1
cp gpurun_out/pmc/summary.txt $O/pmc_summary.txt; cat $O/pmc_summary.txt
