#!/bin/bash
# GPU box: the staggered end of a segment (PMC_DRAIN_KEEP = histories a slot group leaves for every group before it) -> gpurun_out/r06_drain.txt
export TMPDIR=/tmp
mkdir -p gpurun_out
: > gpurun_out/r06_drain.txt
for K in ${KEEPS:-0 1e6 2e6 4e6 8e6 16e6}; do
  PMC_DRAIN_KEEP=$K timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary --no-breakdown --no-counters $BENCH_ARGS > gpurun_out/drain_$K.json 2> gpurun_out/drain_$K.err
  python - $K >> gpurun_out/r06_drain.txt <<'PY'
import json, sys
k = sys.argv[1]
try:
    d = json.loads([l for l in open(f"gpurun_out/drain_{k}.json") if l.startswith("{")][-1])
    print(f"PMC_DRAIN_KEEP {k:>6s}: {d['value']:.4e} packets/s  {d['ms_per_step']:.1f} ms per step  generations {d['roofline']['generations']}")
except Exception as e:
    print(f"PMC_DRAIN_KEEP {k}: failed {e}")
PY
done
cat gpurun_out/r06_drain.txt
