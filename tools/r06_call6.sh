#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config5.py -m gpu -x -q -k "voronoi or cfg5 or config5 or several_components or explicit_absorption or device_memory" > gpurun_out/pytest_voro.log 2>&1; echo "rc $?" >> gpurun_out/pytest_voro.log
tail -5 gpurun_out/pytest_voro.log
for V in new old; do
  if [ $V = old ]; then export PMC_VORO_PLAIN_PROP_ONLY=1; else unset PMC_VORO_PLAIN_PROP_ONLY; fi
  timeout 600 python bench.py --config 5 --store-radiation-field --steps 2 --warmup 1 --packets 2e7 --no-cpu-baseline --no-counters > gpurun_out/c5rf_$V.json 2> gpurun_out/c5rf_$V.err
done
unset PMC_VORO_PLAIN_PROP_ONLY
timeout 600 python bench.py --config 5 --steps 2 --warmup 1 --packets 2e7 --no-cpu-baseline --no-counters > gpurun_out/c5_plain.json 2> gpurun_out/c5_plain.err
for f in c5rf_new c5rf_old c5_plain; do python -c "
import json; d=json.loads([l for l in open('gpurun_out/$f.json') if l.startswith('{')][-1]); print('$f', '%.4g'%d['value'], 'ms %.1f'%d['ms_per_step'])"; done
