#!/bin/bash
# round 3, batch 34 (GPU box): GPU suite on the new defaults (256 slots per claim, 8 steps between round checks); round thresholds again; config 5
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03_batch34; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/gputests.txt 2>&1; echo "gpu tests rc=$?"; grep -E "passed|failed|Error" $O/gputests.txt | tail -3
python tools/sweep.py --packets 1e8 default default libpmc_r32.so libpmc_r48.so libpmc_p4.so libpmc_p16.so libpmc_r32p16.so default > $O/sweep.txt 2>&1
grep -v "amdgpu.ids" $O/sweep.txt | grep "pkt/s" | cut -c1-180
python bench.py --config 5 --steps 2 --warmup 1 --packets 5e7 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('config 5', '%.4g'%d['value'], '%.1f ms'%d['ms_per_step'])"
