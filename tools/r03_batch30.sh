#!/bin/bash
# round 3, batch 30 (GPU box): the peel-off workgroup with unused LDS on top (fewer co-resident workgroups of other kernels)
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03_batch30; mkdir -p $O
python tools/sweep.py --packets 1e8 default default,PMC_PEEL_PAD_LDS=30 default,PMC_PEEL_PAD_LDS=50 default,PMC_PEEL_PAD_LDS=64 default default,PMC_PEEL_PAD_LDS=30 default,PMC_PEEL_PAD_LDS=64 > $O/sweep.txt 2>&1
grep -v "amdgpu.ids" $O/sweep.txt | grep "pkt/s" | cut -c1-180
