#!/bin/bash
# round 3, batch 28 (GPU box): the GPU suite on the 768-lane propagation workgroup; the other workloads
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03_batch28; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/gputests.txt 2>&1; echo "gpu tests rc=$?"; grep -E "passed|failed|Error" $O/gputests.txt | tail -3
for a in "--config 3" "--store-radiation-field" "--source uniform"; do python bench.py $a --steps 2 --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('$a', '%.4g'%d['value'], '%.1f ms'%d['ms_per_step'], 'frac %.4f'%d['roofline']['frac'])"; done
python bench.py --config 4 --steps 2 --warmup 1 --packets 5e7 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('config 4', '%.4g'%d['value'], '%.1f ms'%d['ms_per_step'])"
