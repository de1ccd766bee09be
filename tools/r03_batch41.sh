#!/bin/bash
# round 3, batch 41 (GPU box): radiation-field flavour of the propagation kernel with room for its registers (no scratch)
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03_batch41; mkdir -p $O
for lib in libpmc.so libpmc_rf512.so libpmc_rf256.so libpmc.so; do
PMC_LIBRARY=$R/skirt9_amd/lib/$lib python bench.py --store-radiation-field --steps 2 --warmup 1 --packets 5e7 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('$lib', '%.4g'%d['value'], '%.1f ms'%d['ms_per_step'])"
done
