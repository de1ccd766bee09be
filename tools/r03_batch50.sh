#!/bin/bash
# round 3, batch 50 (GPU box): configs[4] (Voronoi, three instruments, 20 bins, components + statistics): what its transition kernel is made of
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03_batch50; mkdir -p $O
for lib in libpmc.so libpmc_abl_stats.so libpmc_abl_detect.so libpmc_abl_frameadd.so libpmc_abl_hotbins.so; do
(cd /tmp && PMC_LIBRARY=$R/skirt9_amd/lib/$lib PMC_NUM_GROUPS=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$lib -- python $R/bench.py --config 5 --steps 1 --warmup 0 --packets 2e7 --no-cpu-baseline > $O/kt_$lib.log 2>&1)
find $O -name "*kernel_trace.csv" -delete
echo $lib; python - <<PY
import csv,glob
for f in glob.glob("$O/kt_$lib/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n=r["Name"]
        for k in ("transitionKernel","launchKernel","cycleStartKernel","walkKernel","fillBuffer"):
            if k in n: print("   %-20s %8.1f ms per 2e7, %s calls"%(k, float(r["TotalDurationNs"])/1e6, r["Calls"]))
PY
done
