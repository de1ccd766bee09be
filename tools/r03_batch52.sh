#!/bin/bash
# round 3, batch 52 (GPU box): task records without the copies of the slot position and direction -- parity tests, A/B
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03_batch52; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -2
python tools/sweep.py --packets 1e8 libpmc_prev.so default libpmc_prev.so default libpmc_prev.so default > $O/sweep.txt 2>&1; grep "pkt/s" $O/sweep.txt
python tools/sweep.py --packets 1e7 libpmc_prev.so default libpmc_prev.so default > $O/sweep1e7.txt 2>&1; grep "pkt/s" $O/sweep1e7.txt
for lib in libpmc_prev.so libpmc.so; do
(cd /tmp && PMC_LIBRARY=$R/skirt9_amd/lib/$lib PMC_NUM_GROUPS=1 PMC_SERIAL_WALKS=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$lib -- python $R/bench.py --steps 1 --warmup 1 --packets 5e7 --no-cpu-baseline --no-secondary > $O/kt_$lib.log 2>&1)
find $O -name "*kernel_trace.csv" -delete
echo $lib; python - <<PY
import csv,glob
for f in glob.glob("$O/kt_$lib/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n=r["Name"]
        for k in ("transitionKernel","launchKernel","cycleStartKernel","endedScanKernel","walkPropKernel","walkPeelKernel2"):
            if k in n: print("   %-20s %8.1f ms per 1e8"%(k, float(r["TotalDurationNs"])/1e6))
PY
done
