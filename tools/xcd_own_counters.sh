#!/bin/bash
# GPU box: L2 counters (one rocprofv3 --pmc pass) of profiles/microbench/xcd_own.py, and the same command without the profiler -> gpurun_out/xcd_own_l2.txt
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/xcd_own_pmc
rm -rf $O; mkdir -p $O
(cd /tmp && timeout 300 python $R/profiles/microbench/xcd_own.py > $O/plain.log 2>&1)
(cd /tmp && timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum SQ_INSTS_VALU SQ_WAVES --output-format csv -d $O/pass -- python $R/profiles/microbench/xcd_own.py > $O/pass.log 2>&1)
python3 - $O > $R/gpurun_out/xcd_own_l2.txt <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/pass/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "walkOwn" in r["Kernel_Name"]:
            rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], r["Counter_Name"], float(r["Counter_Value"]), r.get("Workgroup_Size"), r.get("Grid_Size")))
by = {}
for d, k, c, v, wg, gs in rows:
    e = by.setdefault(d, {"kernel": k, "wg": wg, "grid": gs})
    e[c] = e.get(c, 0.) + v
for d in sorted(by):
    e = by[d]
    k = e["kernel"]
    mode = "MODE 2 (ownership, free hand-overs)" if ("ILi2E" in k or "<2>" in k) else "MODE 1 (queues)" if ("ILi1E" in k or "<1>" in k) else "MODE 0 (no ownership)"
    req = e.get("TCC_REQ_sum", 0.)
    if req < 1e8: continue   # (warm-up launches)
    print(f"dispatch {d:3d} {mode:38s} workgroup {e['wg']:>5s} grid {e['grid']:>8s}  L2 requests {req:.3e}  miss fraction {e.get('TCC_MISS_sum', 0.) / req:.3f}  VALU {e.get('SQ_INSTS_VALU', 0.):.3e}")
PY
cat $R/gpurun_out/xcd_own_l2.txt; echo "--- without the profiler"; grep "lane-steps/s" $O/plain.log | cut -c1-170; echo "--- under the profiler"; grep "lane-steps/s" $O/pass.log | cut -c1-170
