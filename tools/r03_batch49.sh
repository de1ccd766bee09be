#!/bin/bash
# round 3, batch 49 (GPU box): kernel trace of configs[4] (Voronoi)
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03_batch49; mkdir -p $O
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python $R/bench.py --config 5 --steps 1 --warmup 1 --packets 2e7 --no-cpu-baseline > $O/kt.log 2>&1)
find $O -name "*kernel_trace.csv" -delete
python - <<PY
import csv,glob
for f in glob.glob("$O/kt/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:8]:
        print("   %-60s calls %5s total %8.1f ms avg %8.1f us %5s %%"%(r["Name"][:60], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3, r["Percentage"]))
PY
grep "^{" $O/kt.log | tail -1 | cut -c1-300
