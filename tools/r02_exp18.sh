#!/bin/bash
# round 2, experiment 18: what the launch / transition kernels spend their time on (ablations, section timers)
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r02_exp18; rm -rf $O; mkdir -p $O
for v in nostats nodetect; do
(cd /tmp && PMC_LIBRARY=$R/skirt9_amd/lib/libpmc_$v.so PMC_NUM_GROUPS=1 PMC_SERIAL_WALKS=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$v -- python $R/bench.py --steps 1 --warmup 0 --packets 5e7 --no-cpu-baseline > $O/kt_$v.log 2>&1)
echo "== $v"; find $O/kt_$v -name "*kernel_stats.csv" -exec cat {} \; | head -6 | cut -c1-130
done
find $O -name "*kernel_trace.csv" -size +20M -delete
timeout 600 python tools/sweep.py --packets 5e7 libpmc_prof.so,PMC_NUM_GROUPS=1,PMC_SERIAL_WALKS=1,PMC_PROFILE_DUMP=1 > $O/sweep.txt 2> $O/sweep.err
grep "PMC_PROFILE transition\|PMC_PROFILE launch" $O/sweep.err
