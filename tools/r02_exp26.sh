#!/bin/bash
# round 2, experiment 26: launch kernel with the ended slots of a tile compacted (full waves)
export TMPDIR=/tmp
O=gpurun_out/r02_exp26; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log
timeout 900 python tools/sweep.py --packets 5e7 default default,PMC_NUM_GROUPS=1,PMC_SERIAL_WALKS=1,PMC_TIMING_DUMP=1 > $O/sweep.txt 2> $O/sweep.err
cut -c1-170 $O/sweep.txt; grep PMC_TIMING $O/sweep.err | tail -1
timeout 1200 python bench.py --config 5 --steps 2 --warmup 1 --packets 5e7 --no-cpu-baseline > $O/bench_config5.json 2> $O/c5.err; cut -c1-200 $O/bench_config5.json
