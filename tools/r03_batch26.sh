#!/bin/bash
# round 3, batch 26 (GPU box): around the 768-lane propagation workgroup: pass-1 records, peel-off workgroup size, round threshold, slot groups
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03_batch26; mkdir -p $O
python tools/sweep.py --packets 1e8 default libpmc_b768t4.so libpmc_b768t5.so libpmc_b768t4p768.so libpmc_b768t4p1024.so libpmc_b768t4r32.so libpmc_b768t4r48.so \
   libpmc_b768t4.so,PMC_NUM_GROUPS=2 libpmc_b768t4.so,PMC_NUM_GROUPS=4 libpmc_b768t4.so,PMC_NUM_SLOTS=12582912 libpmc_b768t4.so,PMC_NUM_SLOTS=6291456 libpmc_b768t4.so default > $O/sweep.txt 2>&1
grep -v "amdgpu.ids" $O/sweep.txt | grep "pkt/s" | cut -c1-160
