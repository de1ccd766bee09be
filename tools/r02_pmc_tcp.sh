#!/bin/bash
# TA / TCP counter passes over one bench step (serial walk kernels, one slot group), for the peel-off kernel forms
export TMPDIR=/tmp
R=$PWD
export PMC_NUM_GROUPS=1 PMC_SERIAL_WALKS=1 PMC_WALK_BLOCKS_PER_CU=3
PASSES=(
"TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum"
"TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum"
"TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"
"SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD"
"SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA"
)
for form in v1 v3; do
  OUT=$R/gpurun_out/pmc_$form
  rm -rf $OUT; mkdir -p $OUT
  if [ $form = v1 ]; then export PMC_PEEL_V1=1; else unset PMC_PEEL_V1; fi
  i=0
  for p in "${PASSES[@]}"; do
    i=$((i+1))
    (cd /tmp && timeout 120 rocprofv3 --pmc $p --output-format csv -d $OUT/pass$i -- python $R/bench.py --steps 1 --warmup 0 --packets 2e7 --no-cpu-baseline --no-secondary > $OUT/pass$i.log 2>&1)
  done
  OUTDIR=$OUT python3 - <<'PY'
import csv, glob, collections, os, re
out = os.environ["OUTDIR"]
tot = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(out + "/pass*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        m = re.search(r"(walkPeelKernel2|walkPeelKernel|walkPropKernel|transitionKernel|launchKernel)", row["Kernel_Name"])
        if not m: continue
        tot[m.group(1)][row["Counter_Name"]] += float(row["Counter_Value"])
with open(out + "/summary.txt", "w") as fh:
    for k in sorted(tot):
        fh.write(f"== {k}\n")
        for c in sorted(tot[k]):
            fh.write(f"   {c:45s} {tot[k][c]:.6e}\n")
PY
  find $OUT -name "*.csv" -size +1M -delete
done
grep -A40 "walkPeel" gpurun_out/pmc_v1/summary.txt | head -45; grep -A40 "walkPeel" gpurun_out/pmc_v3/summary.txt | head -45
