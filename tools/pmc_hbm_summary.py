#!/usr/bin/env python3
"""Sums the FETCH_SIZE / WRITE_SIZE counter rows of the separate rocprofv3 --pmc passes per kernel.
usage: tools/pmc_hbm_summary.py gpurun_out/<run> > profiles/<run>_pmc_hbm.csv
(the passes are made by tools/run_profile_set.sh: one bench step of 2e7 packets each)"""
import collections
import csv
import glob
import re
import sys

run = sys.argv[1]
print("counter,kernel,launches,sum_KiB_per_step_of_2e7_packets")
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    tot = collections.defaultdict(float)
    calls = collections.defaultdict(set)
    for f in glob.glob(f"{run}/pmc_{counter}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            m = re.search(r"(voroPropKernel|voroPeelKernel|walkPeelKernel|walkPropKernel|walkKernel|transitionKernel|launchKernel|cycleStartKernel|endedScanKernel|statMergeKernel|peelSortCountKernel|peelSortOffsetsKernel|peelSortScatterKernel|rfHistKernel|rfScanKernel|rfScatterKernel|rfReduceKernel)", row["Kernel_Name"])
            if not m or row["Counter_Name"] != counter:
                continue
            tot[m.group(1)] += float(row["Counter_Value"])
            calls[m.group(1)].add(row["Dispatch_Id"])
    for k in tot:
        print(f"{counter},{k},{len(calls[k])},{tot[k]:.6e}")
