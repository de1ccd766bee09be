#!/bin/bash
# Tuning aid (GPU box): A/B runs of engine variants on one box, one scene set-up.
#   tools/ab.sh OUTNAME [--packets N] [--ski FILE] VARIANT...     (VARIANT as in tools/sweep.py: lib[,ENV=VALUE...])
# Build the variants first, here in the container:  make variant NAME=foo DEFS=-DPMC_FOO=1   (-> skirt9_amd/lib/libpmc_foo.so)
# Output: gpurun_out/OUTNAME.txt (copy what is worth keeping to profiles/sweeps/).
# "serial" as a variant suffix is shorthand for one slot group with the walk kernels in series and per-kernel HIP-event times.
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/$1.txt; shift
mkdir -p $R/gpurun_out
ARGS=()
for a in "$@"; do
  ARGS+=("${a//,serial/,PMC_NUM_GROUPS=1,PMC_SERIAL_WALKS=1,PMC_TIMING_DUMP=1}")
done
python tools/sweep.py "${ARGS[@]}" > $OUT 2>&1
grep -A1 "pkt/s\|PMC_PROFILE\|PMC_TIMING\|Error\|error" $OUT | grep -v "^--" | cut -c1-400
