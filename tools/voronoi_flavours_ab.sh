#!/bin/bash
# GPU box: configs[4] with radiation field / explicit absorption in the Voronoi kernels (new) against the generic kernel (old) -> gpurun_out/r06_voro_flavours.txt
export TMPDIR=/tmp
mkdir -p gpurun_out
: > gpurun_out/r06_voro_flavours.txt
for F in "--store-radiation-field" "--explicit-absorption" "--store-radiation-field --explicit-absorption"; do
for V in new old; do
  if [ $V = old ]; then export PMC_VORO_PLAIN_PROP_ONLY=1; else unset PMC_VORO_PLAIN_PROP_ONLY; fi
  timeout 600 python bench.py --config 5 $F --steps 2 --warmup 1 --packets 2e7 --no-cpu-baseline --no-counters > gpurun_out/c5f.json 2> gpurun_out/c5f.err
  python -c "
import json; d=json.loads([l for l in open('gpurun_out/c5f.json') if l.startswith('{')][-1]); print('config 5 $F [$V]: %.4g packets/s, %.1f ms per step of 2e7'%(d['value'], d['ms_per_step']))" >> gpurun_out/r06_voro_flavours.txt 2>&1
done; done
cat gpurun_out/r06_voro_flavours.txt
