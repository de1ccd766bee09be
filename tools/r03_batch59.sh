#!/bin/bash
# round 3, batch 59 (GPU box): Voronoi exit search over 16-byte float entries (PMC_VORO_COMPACT=1): bit-exact ray tests, parity, configs[4]
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03_batch59; mkdir -p $O
PMC_VORO_COMPACT=1 timeout 900 python -m pytest tests -m gpu -x -q -k "voronoi or cfg5 or config5 or Voronoi" > $O/pytest.log 2>&1; grep -E "passed|failed|Error" $O/pytest.log | tail -3
for c in 0 1 0 1; do
PMC_VORO_COMPACT=$c python bench.py --config 5 --steps 2 --warmup 1 --packets 2e7 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('compact=$c', '%.4g'%d['value'], '%.1f ms'%d['ms_per_step'], 'visits/packet %.1f'%d['roofline']['cell_visits_per_packet'])"
done
