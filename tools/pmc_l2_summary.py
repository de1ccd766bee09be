#!/usr/bin/env python3
"""Sums the L2 / VALU counter rows of ONE rocprofv3 --pmc pass per kernel.
usage: tools/pmc_l2_summary.py gpurun_out/<run>/pmc_l2 > profiles/<run>_pmc_l2.csv
(the pass is made by tools/run_profile_set.sh over one bench step of 2e7 packets without the one-group breakdown:
 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum SQ_INSTS_VALU SQ_WAVES)"""
import collections
import csv
import glob
import re
import sys

run = sys.argv[1]
names = ("TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum", "SQ_INSTS_VALU", "SQ_WAVES")
tot = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(set)
for f in glob.glob(f"{run}/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        m = re.search(r"(voroPropKernel|voroPeelKernel|walkPeelKernel|walkPropKernel|walkKernel|transitionKernel|launchKernel|cycleStartKernel|endedScanKernel|statMergeKernel|peelSortCountKernel|peelSortOffsetsKernel|rfHistKernel|rfScanKernel|rfScatterKernel|rfReduceKernel|statReduceKernel)", row["Kernel_Name"])
        if not m or row["Counter_Name"] not in names:
            continue
        tot[m.group(1)][row["Counter_Name"]] += float(row["Counter_Value"])
        calls[m.group(1)].add(row["Dispatch_Id"])
print("kernel,launches," + ",".join(n + "_per_step_of_2e7_packets" for n in names))
for k in sorted(tot):
    print(f"{k},{len(calls[k])}," + ",".join(f"{tot[k][n]:.6e}" for n in names))
