#!/bin/bash
# round 3, batch 4 (GPU box): GPU tests after the statistics pool and the per-slot dust properties; bench lines of configs 2 and 3
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03_batch4; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/gputests.txt 2>&1; echo "gpu tests rc=$?"; tail -15 $O/gputests.txt | cut -c1-300
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python -c "
import json;d=json.load(open('$O/bench.json'));print(d['value'],d['ms_per_step'],{k:(v if not isinstance(v,dict) else {a:(round(b,4) if isinstance(b,float) else b) for a,b in v.items()}) for k,v in d['roofline'].items() if k in ('frac','gather_frac','prop','peel','segment_ms','transition_kernel_ms','generations')})"
python bench.py --config 3 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_config3.json 2> $O/bench3.err; echo "bench3 rc=$?"; python -c "
import json;d=json.load(open('$O/bench_config3.json'));print(d['value'],d['ms_per_step'])"
