#!/bin/bash
# tuning aid (GPU box): hardware-counter passes over one bench step, summed per kernel -> gpurun_out/pmc/summary.txt
# usage: tools/pmc_passes.sh [bench args...]
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc
rm -rf $OUT; mkdir -p $OUT
ARGS="${@:---steps 1 --warmup 0 --packets 2e7 --no-cpu-baseline --no-counters}"
PASSES=(
"SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD"
"SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
"SQ_WAVES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES"
"TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE"
)
i=0
for p in "${PASSES[@]}"; do
  i=$((i+1))
  (cd /tmp && timeout 240 rocprofv3 --pmc $p --output-format csv -d $OUT/pass$i -- python $OLDPWD/bench.py $ARGS > $OUT/pass$i.log 2>&1)
done
python3 - <<'PY'
import csv, glob, collections, os
out = os.environ.get("OUTDIR", os.getcwd() + "/gpurun_out/pmc")
tot = collections.defaultdict(lambda: collections.defaultdict(float))
passes = collections.defaultdict(lambda: collections.defaultdict(set))
calls = collections.defaultdict(set)
for f in glob.glob(out + "/pass*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        import re
        m = re.search(r"(walkPeelKernel|walkPropKernel|walkKernel|transitionKernel|launchKernel|cycleStartKernel|endedScanKernel|statMergeKernel|chase)", row["Kernel_Name"])
        if not m: continue
        k = m.group(1)
        tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
        passes[k][row["Counter_Name"]].add(f)
        calls[k].add((f, row["Dispatch_Id"]))
for k in tot:
    for c in tot[k]:
        tot[k][c] /= len(passes[k][c])  # a counter listed in several passes: the mean over them
with open(out + "/summary.txt", "w") as fh:
    for k in sorted(tot):
        fh.write(f"== {k}\n")
        for c in sorted(tot[k]):
            fh.write(f"   {c:45s} {tot[k][c]:.6e}\n")
print(open(out + "/summary.txt").read())
PY
