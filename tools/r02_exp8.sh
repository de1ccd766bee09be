#!/bin/bash
# round 2, experiment 8: calibration of the SQ VALU counters on kernels of known instruction counts (valu_rate2)
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r02_exp8; rm -rf $O; mkdir -p $O
(cd /tmp && timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU --output-format csv -d $O/pmc -- $R/profiles/microbench/valu_rate2 > $O/run.log 2>&1)
python3 - <<'PY'
import csv, glob, collections, os
out = os.environ.get("O", os.getcwd() + "/gpurun_out/r02_exp8")
rows = collections.OrderedDict()
for f in glob.glob(os.getcwd() + "/gpurun_out/r02_exp8/pmc/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        key = (int(row["Dispatch_Id"]), row["Kernel_Name"][:40], row["Grid_Size"])
        rows.setdefault(key, {})[row["Counter_Name"]] = float(row["Counter_Value"])
for k in sorted(rows):
    v = rows[k]
    print(k, " ".join(f"{c}={v[c]:.4e}" for c in sorted(v)))
PY
