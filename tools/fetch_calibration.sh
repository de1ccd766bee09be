#!/bin/bash
# GPU box: what does one L2 miss move, and what does FETCH_SIZE count for it?  (round-5 review, item 4)  -> gpurun_out/fetch_calibration.txt
# counters of the chaseTrue launches of profiles/microbench/fetch_calib.py (a known number of random gathers from a 256 MB table), per gather
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/fetch_calib
rm -rf $O; mkdir -p $O
STEPS=1000
i=0
for P in "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_MISS_sum TCC_REQ_sum" "TCC_BUBBLE_sum TCC_EA0_RD_UNCACHED_32B_sum TCC_HIT_sum TCC_READ_sum"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --pmc $P --output-format csv -d $O/pass$i -- python $R/profiles/microbench/fetch_calib.py $STEPS > $O/pass$i.log 2>&1)
done
# wide coalesced streams and scattered 8-byte stores of a known size (2 GiB; 2^25 sectors)
for P in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  i=$((i+1))
  (cd /tmp && FETCH_CALIB_STREAMS=1 timeout 200 rocprofv3 --pmc $P --output-format csv -d $O/stream$i -- python $R/profiles/microbench/fetch_calib.py > $O/stream$i.log 2>&1)
done
python3 - $O $STEPS > $R/gpurun_out/fetch_calibration.txt <<'PY'
import csv, glob, collections, sys
out, steps = sys.argv[1], int(sys.argv[2])
lanes = 256 * 768
rows = collections.defaultdict(dict)   # dispatch order -> counters
for f in sorted(glob.glob(out + "/pass*/**/*counter_collection.csv", recursive=True)):
    ks = [r for r in csv.DictReader(open(f)) if "chaseTrue" in r["Kernel_Name"]]
    ids = sorted({int(r["Dispatch_Id"]) for r in ks})
    for r in ks:
        rows[ids.index(int(r["Dispatch_Id"]))][r["Counter_Name"]] = rows[ids.index(int(r["Dispatch_Id"]))].get(r["Counter_Name"], 0.) + float(r["Counter_Value"])
names = ["32-byte records, 256 MB table: warm-up", "32-byte records, 256 MB table", "16-byte records, 256 MB table: warm-up", "16-byte records, 256 MB table",
         "32-byte records, 32 MB table: warm-up", "32-byte records, 32 MB table"]
for i in sorted(rows):
    n = lanes * (10 if i % 2 == 0 else steps)
    print(f"{names[i] if i < len(names) else i}: {n} gathers")
    for c, v in sorted(rows[i].items()):
        extra = f"  = {v * 1024 / n:8.2f} B per gather if the unit is KiB" if c == "FETCH_SIZE" else ""
        print(f"    {c:28s} {v:.6e}  {v / n:8.4f} per gather{extra}")
rows = collections.defaultdict(dict)
for f in sorted(glob.glob(out + "/stream*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        for k in ("calibStreamRead", "calibStreamWrite", "calibScatterWrite"):
            if k in r["Kernel_Name"]:
                rows[k][r["Counter_Name"]] = rows[k].get(r["Counter_Name"], 0.) + float(r["Counter_Value"])
nbytes = 1 << 31
for k, what, n in (("calibStreamRead", "bytes read (16 B per lane, coalesced)", nbytes), ("calibStreamWrite", "bytes written (16 B per lane, coalesced)", nbytes),
                   ("calibScatterWrite", "8-byte stores, each into a 64-byte sector of its own", nbytes // 64)):
    print(f"{k}: {n} {what}")
    for c, v in sorted(rows[k].items()):
        print(f"    {c:28s} {v:.6e}  {v / n:10.6f} per unit" + (f"  = {v * 1024 / n:8.3f} B per unit if the counter is in KiB" if c in ("FETCH_SIZE", "WRITE_SIZE") else ""))
PY
cat $R/gpurun_out/fetch_calibration.txt
