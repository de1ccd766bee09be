#!/bin/bash
# round 3, batch 56 (GPU box): one-wavelength scenes: wavelength bin per instrument as a constant; no head-record read for emission peel-offs -- parity tests, A/B
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03_batch56; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -2
PMC_NO_MONO=1 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "cfg1 or cfg2small" > $O/pytest_nomono.log 2>&1; grep -E "passed|failed" $O/pytest_nomono.log | tail -2
python tools/sweep.py --packets 1e8 libpmc_prev.so default libpmc_prev.so default default,PMC_NO_MONO=1 libpmc_prev.so default > $O/sweep.txt 2>&1; grep "pkt/s" $O/sweep.txt
