#!/bin/bash
# round 3, batch 35 (GPU box): stream priorities
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03_batch35; mkdir -p $O
python tools/sweep.py --packets 1e8 default default default,PMC_STREAM_PRIORITY=1 default,PMC_STREAM_PRIORITY=2 default default,PMC_STREAM_PRIORITY=1 default,PMC_STREAM_PRIORITY=2 > $O/sweep.txt 2>&1
grep -v "amdgpu.ids" $O/sweep.txt | grep "pkt/s" | cut -c1-180
