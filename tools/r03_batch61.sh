#!/bin/bash
# round 3, batch 61 (GPU box): cell gather addressed by a 32-bit byte offset from a scalar base -- ray tests, A/B
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03_batch61; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -2
python tools/sweep.py --packets 1e8 libpmc_prev.so default libpmc_prev.so default libpmc_prev.so default libpmc_prev.so default > $O/sweep.txt 2>&1; grep "pkt/s" $O/sweep.txt
