#!/bin/bash
# round 3, batch 13 (GPU box): is the launch kernel bound by its scattered stores?  (seven more per history)
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03_batch13; mkdir -p $O
for lib in libpmc.so libpmc_lst.so; do
(cd /tmp && PMC_LIBRARY=$R/skirt9_amd/lib/$lib PMC_NUM_GROUPS=1 PMC_SERIAL_WALKS=1 PMC_WALK_BLOCKS_PER_CU=3 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$lib -- python $R/bench.py --steps 1 --warmup 1 --packets 5e7 --no-cpu-baseline --no-secondary > $O/kt_$lib.log 2>&1)
f=$(find $O/kt_$lib -name "*kernel_stats.csv" | head -1); echo $lib; cut -d, -f1-4 $f | grep -i "launchKernel\|transitionKernel\|cycleStart" | cut -c1-120
find $O -name "*kernel_trace.csv" -delete
done
