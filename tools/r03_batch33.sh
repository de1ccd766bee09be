#!/bin/bash
# round 3, batch 33 (GPU box): slots per claim x steps between round checks
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03_batch33; mkdir -p $O
python tools/sweep.py --packets 1e8 default libpmc_a.so libpmc_b.so libpmc_c.so libpmc_d.so libpmc_e.so libpmc_f.so default libpmc_a.so libpmc_b.so libpmc_c.so libpmc_d.so libpmc_e.so > $O/sweep.txt 2>&1
grep -v "amdgpu.ids" $O/sweep.txt | grep "pkt/s" | cut -c1-180
