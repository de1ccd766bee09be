#!/bin/bash
# round 3, batch 20 (GPU box): sixteen instruments / sources: the whole GPU suite, A/B of the headline
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03_batch20; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/gputests.txt 2>&1; echo "gpu tests rc=$?"; tail -4 $O/gputests.txt | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python tools/sweep.py --packets 1e8 libpmc_r02.so default libpmc_r02.so default > $O/sweep.txt 2>&1
grep -v "amdgpu.ids" $O/sweep.txt | grep "pkt/s" | cut -c1-160
