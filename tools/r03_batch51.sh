#!/bin/bash
# round 3, batch 51 (GPU box): Voronoi cycle start kernel with rounds of neighbours in flight (as the walk kernel); parity of the Voronoi scenes
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03_batch51; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "voronoi or cfg5 or config5" > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -2
for lib in libpmc_vs1.so libpmc.so libpmc_vsw3.so libpmc_vsw4.so libpmc_vs1.so libpmc.so; do
PMC_LIBRARY=$R/skirt9_amd/lib/$lib python bench.py --config 5 --steps 2 --warmup 1 --packets 2e7 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('$lib', '%.4g'%d['value'], '%.1f ms'%d['ms_per_step'])"
done
