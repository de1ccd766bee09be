#!/bin/bash
# round 3, batch 60 (GPU box): Voronoi exit search over 16-byte float entries, 8 / 12 / 16 entries in flight
export TMPDIR=/tmp
R=$PWD
for v in "libpmc.so 0" "libpmc_vc8.so 1" "libpmc_vc12.so 1" "libpmc_vc16.so 1"; do
set -- $v
PMC_LIBRARY=$R/skirt9_amd/lib/$1 PMC_VORO_COMPACT=$2 python bench.py --config 5 --steps 2 --warmup 1 --packets 2e7 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('$1 compact=$2', '%.4g'%d['value'], '%.1f ms'%d['ms_per_step'])"
done
