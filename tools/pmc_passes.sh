#!/bin/bash
# tuning aid (GPU box): hardware-counter passes over one bench step, summed per kernel -> gpurun_out/pmc/summary.txt
# usage: tools/pmc_passes.sh [bench args...]
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc
rm -rf $OUT; mkdir -p $OUT
ARGS="${@:---steps 1 --warmup 0 --packets 2e7 --no-cpu-baseline}"
PASSES=(
"SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_SMEM"
"SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INST_CYCLES_SMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES"
)
i=0
for p in "${PASSES[@]}"; do
  i=$((i+1))
  (cd /tmp && timeout 90 rocprofv3 --pmc $p --output-format csv -d $OUT/pass$i -- python $OLDPWD/bench.py $ARGS > $OUT/pass$i.log 2>&1)
done
python3 - <<'PY'
import csv, glob, collections, os
out = os.environ.get("OUTDIR", os.getcwd() + "/gpurun_out/pmc")
tot = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(set)
for f in glob.glob(out + "/pass*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        import re
        m = re.search(r"(walkKernel|transitionKernel|launchKernel|chase)", row["Kernel_Name"])
        if not m: continue
        k = m.group(1)
        tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
        calls[k].add((f, row["Dispatch_Id"]))
with open(out + "/summary.txt", "w") as fh:
    for k in sorted(tot):
        fh.write(f"== {k}\n")
        for c in sorted(tot[k]):
            fh.write(f"   {c:45s} {tot[k][c]:.6e}\n")
print(open(out + "/summary.txt").read())
PY
