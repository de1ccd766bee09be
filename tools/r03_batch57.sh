#!/bin/bash
# round 3, batch 57 (GPU box): launch kernel with every other wave starting 10 / 20 / 40 us late (flush atomics of one wave beside the sampling arithmetic of another?)
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03_batch57; mkdir -p $O
for lib in libpmc.so libpmc_stag3.so libpmc_stag6.so libpmc_stag12.so libpmc.so; do
(cd /tmp && PMC_LIBRARY=$R/skirt9_amd/lib/$lib PMC_NUM_GROUPS=1 PMC_SERIAL_WALKS=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$lib -- python $R/bench.py --steps 1 --warmup 1 --packets 5e7 --no-cpu-baseline --no-secondary > $O/kt_$lib.log 2>&1)
find $O -name "*kernel_trace.csv" -delete
echo $lib; python - <<PY
import csv,glob
for f in glob.glob("$O/kt_$lib/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n=r["Name"]
        for k in ("transitionKernel","launchKernel","cycleStartKernel"):
            if k in n: print("   %-20s %8.1f ms per 1e8"%(k, float(r["TotalDurationNs"])/1e6))
PY
rm -rf $O/kt_$lib
done
