#!/bin/bash
# round 3, batch 23 (GPU box): Cartesian grids beyond 1024 cells per axis, deep octrees after the constant folding: the GPU suite
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03_batch23; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/gputests.txt 2>&1; echo "gpu tests rc=$?"; grep -E "passed|failed|Error" $O/gputests.txt | tail -3
python tools/sweep.py --packets 1e8 libpmc_prev.so default libpmc_prev.so default > $O/sweep.txt 2>&1
grep -v "amdgpu.ids" $O/sweep.txt | grep "pkt/s" | cut -c1-160
