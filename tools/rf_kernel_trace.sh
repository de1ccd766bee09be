#!/bin/bash
# GPU box: kernel trace of the radiation-field workload, one slot group, walk kernels in series, 2e7 packets -> the table on stdout
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/rf_kt6
rm -rf $O; mkdir -p $O
(cd /tmp && PMC_NUM_GROUPS=1 PMC_SERIAL_WALKS=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python $R/bench.py --store-radiation-field --steps 1 --warmup 0 --packets 2e7 --no-cpu-baseline --no-secondary --no-breakdown --no-counters > $O/kt.log 2>&1)
python3 - $O <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/kt/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:14]:
        print(f"{r['Name'][:90]:90s} calls {r['Calls']:>5s} total_ms {float(r['TotalDurationNs'])/1e6:9.2f} avg_us {float(r['AverageNs'])/1e3:9.1f} {r['Percentage']:>6s}%")
PY
find $O -name "*kernel_trace.csv" -delete
