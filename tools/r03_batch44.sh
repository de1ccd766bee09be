#!/bin/bash
# round 3, batch 44 (GPU box): sparse generations: transition and cycle start kernels over the lists too -- parity tests, A/B against the previous commit, the generations of the drain
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03_batch44; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -2
python tools/sweep.py --packets 1e8 libpmc_prev.so default libpmc_prev.so default default,PMC_NO_LIVE_LISTS=1 > $O/sweep.txt 2>&1; grep "pkt/s" $O/sweep.txt
python tools/sweep.py --packets 1e7 libpmc_prev.so default libpmc_prev.so default > $O/sweep1e7.txt 2>&1; grep "pkt/s" $O/sweep1e7.txt
PMC_GEN_DUMP=1 python tools/sweep.py --packets 1e8 default > $O/gens.txt 2>&1
grep PMC_GEN $O/gens.txt | tail -34
