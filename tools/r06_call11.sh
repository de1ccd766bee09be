#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
BENCH_SHARE_DEVICE=1 timeout 900 python bench.py --gpus 8 --steps 2 --warmup 1 --packets 1e7 --no-cpu-baseline --no-secondary --no-breakdown --no-counters > gpurun_out/share8.json 2> gpurun_out/share8.err; echo "rc $?"
tail -c 1800 gpurun_out/share8.json; echo; grep -i "libpmc\|error\|Traceback" gpurun_out/share8.err | head -20
