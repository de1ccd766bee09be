#!/bin/bash
# round 2, experiment 10: per-dispatch VALU counters of the walk kernels (default and variants), one slot group, serial walks
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r02_exp10; rm -rf $O; mkdir -p $O
for v in default s8 r32; do
  lib=$R/skirt9_amd/lib/libpmc.so; [ $v != default ] && lib=$R/skirt9_amd/lib/libpmc_$v.so
  (cd /tmp && PMC_LIBRARY=$lib PMC_NUM_GROUPS=1 PMC_SERIAL_WALKS=1 timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $O/$v -- python $R/bench.py --steps 1 --warmup 0 --packets 2e7 --no-cpu-baseline > $O/$v.log 2>&1)
done
python3 - <<'PY'
import csv, glob, collections, os, re
base = os.getcwd() + "/gpurun_out/r02_exp10"
for v in ("default", "s8", "r32"):
    rows = collections.OrderedDict()
    for f in glob.glob(f"{base}/{v}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            m = re.search(r"(walkPeelKernel|walkPropKernel)", row["Kernel_Name"])
            if not m: continue
            rows.setdefault((m.group(1), int(row["Dispatch_Id"])), {})[row["Counter_Name"]] = float(row["Counter_Value"])
    for kern in ("walkPeelKernel", "walkPropKernel"):
        ks = sorted(k for k in rows if k[0] == kern)
        print(f"== {v} {kern}: {len(ks)} dispatches")
        tot = collections.defaultdict(float)
        for i, k in enumerate(ks):
            r = rows[k]
            for c in r: tot[c] += r[c]
            if i < 6 or i % 4 == 0:
                print(f"  #{i:2d} VALU {r['SQ_INSTS_VALU']:.3e} lanes/instr {r['SQ_THREAD_CYCLES_VALU']/r['SQ_INSTS_VALU']:5.1f} active/instr {r['SQ_ACTIVE_INST_VALU']/r['SQ_INSTS_VALU']:.2f} SALU {r['SQ_INSTS_SALU']:.3e} wavecyc {r['SQ_WAVE_CYCLES']:.3e} wait {r['SQ_WAIT_ANY']/r['SQ_WAVE_CYCLES']:.2f} busy {r['SQ_BUSY_CYCLES']:.3e} waves {r['SQ_WAVES']:.0f}")
        print("  total", " ".join(f"{c}={tot[c]:.4e}" for c in sorted(tot)))
PY
