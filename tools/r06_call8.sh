#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "rc $?" >> gpurun_out/pytest_gpu.log
tail -15 gpurun_out/pytest_gpu.log
timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-secondary --no-counters > gpurun_out/bench_deep.json 2> gpurun_out/bench_deep.err
python -c "
import json; d=json.loads([l for l in open('gpurun_out/bench_deep.json') if l.startswith('{')][-1]); print('headline %.4g packets/s, %.1f ms per step'%(d['value'], d['ms_per_step'])); print(d['roofline']['serial_kernel_ms_per_step'])"
