#!/bin/bash
# round 3, batch 22 (GPU box): octrees of 13-15 levels (coordinate table in global memory): parity; A/B of the headline against the previous commit
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03_batch22; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -m gpu -x -q > $O/gputests.txt 2>&1; echo "gpu tests rc=$?"; tail -5 $O/gputests.txt | cut -c1-300
python tools/sweep.py --packets 1e8 libpmc_prev.so default libpmc_prev.so default libpmc_prev.so default > $O/sweep.txt 2>&1
grep -v "amdgpu.ids" $O/sweep.txt | grep "pkt/s" | cut -c1-160
python tools/sweep.py --packets 2e7 --ski tests/ski/cfg2deep.ski libpmc_prev.so default libpmc_prev.so default > $O/sweep_deep.txt 2>&1
grep -v "amdgpu.ids" $O/sweep_deep.txt | grep "pkt/s" | cut -c1-160
python tools/sweep.py --packets 2e7 --ski tests/ski/cfg2deeper.ski default default > $O/sweep_deeper.txt 2>&1
grep -v "amdgpu.ids" $O/sweep_deeper.txt | grep "pkt/s" | cut -c1-160
