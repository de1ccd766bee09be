#!/bin/bash
# round 3, batch 48 (GPU box): what the source sampling of the launch kernel is made of (ablation builds: wrong results, tuning only)
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03_batch48; mkdir -p $O
for lib in libpmc.so libpmc_abl_dir.so libpmc_abl_loglog.so libpmc_abl_rng.so libpmc_abl_flush.so; do
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
done
