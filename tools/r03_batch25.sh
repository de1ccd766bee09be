#!/bin/bash
# round 3, batch 25 (GPU box): the propagation kernel as ONE larger workgroup per CU (512 / 768 / 1024 lanes sharing one coordinate table)
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03_batch25; mkdir -p $O
python tools/sweep.py --packets 1e8 default libpmc_b768t2.so libpmc_b768t3.so libpmc_b768t4.so libpmc_b512t4.so libpmc_b512t6.so libpmc_b1024t2.so default libpmc_b768t2.so libpmc_b768t4.so libpmc_b512t4.so libpmc_b1024t2.so > $O/sweep.txt 2>&1
grep -v "amdgpu.ids" $O/sweep.txt | grep "pkt/s" | cut -c1-160
