#!/bin/bash
# tuning aid (GPU box): ONE rocprofv3 counter pass over one bench step of 2e7 packets (Sersic source only), summed per kernel
# (PMC_PASS_ARGS: further bench.py arguments, e.g. "--config 5")
# usage: tools/pmc_pass.sh NAME "COUNTER COUNTER ..." [engine library under skirt9_amd/lib]     -> gpurun_out/NAME.txt
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/pmc_$1
rm -rf $OUT; mkdir -p $OUT
[ -n "$3" ] && export PMC_LIBRARY=$R/skirt9_amd/lib/$3
(cd /tmp && timeout 400 rocprofv3 --pmc $2 --output-format csv -d $OUT/pass -- python $R/bench.py --steps 1 --warmup 0 --packets 2e7 --no-cpu-baseline --no-secondary --no-counters $PMC_PASS_ARGS > $OUT/pass.log 2>&1)
python3 - $OUT > $R/gpurun_out/$1.txt <<'PY'
import csv, glob, collections, re, sys
tot = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(sys.argv[1] + "/pass/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        m = re.search(r"(voroPropKernel|voroPeelKernel|walkPeelKernel|walkPropKernel|walkKernel|transitionKernel|launchKernel|cycleStartKernel|endedScanKernel|statMergeKernel)", row["Kernel_Name"])
        if m: tot[m.group(1)][row["Counter_Name"]] += float(row["Counter_Value"])
for k in sorted(tot):
    print(f"{k:18s} " + "  ".join(f"{c} {v:.4e}" for c, v in sorted(tot[k].items())))
PY
grep "walk\|voro" $R/gpurun_out/$1.txt
rm -rf $OUT
