#!/bin/bash
# round 3, batch 19 (GPU box): live slots and kernel times of every generation of a 1e8-packet segment (the drain); parity of cfg2nf
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03_batch19; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "cfg2nf or cfg3file" > $O/gputests.txt 2>&1; echo "gpu tests rc=$?"; tail -3 $O/gputests.txt | cut -c1-200
PMC_GEN_DUMP=1 python tools/sweep.py --packets 1e8 default > $O/gens.txt 2>&1
grep PMC_GEN $O/gens.txt | tail -60 | head -5; grep -c PMC_GEN $O/gens.txt
