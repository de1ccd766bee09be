#!/bin/bash
# round 3, batch 47 (GPU box): the scan of the ended-history counts as the work of the transition kernel's last workgroup -- parity tests, A/B
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03_batch47; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -2
python tools/sweep.py --packets 1e8 libpmc_prev.so default libpmc_prev.so default libpmc_prev.so default > $O/sweep.txt 2>&1; grep "pkt/s" $O/sweep.txt
python tools/sweep.py --packets 1e7 libpmc_prev.so default libpmc_prev.so default > $O/sweep1e7.txt 2>&1; grep "pkt/s" $O/sweep1e7.txt
