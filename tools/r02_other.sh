#!/bin/bash
# round 2: the other workloads of the same path on the current build (bench lines under gpurun_out/r02_other)
export TMPDIR=/tmp
O=gpurun_out/${RUN_NAME:-r02_other}; rm -rf $O; mkdir -p $O
timeout 900 python bench.py --config 3 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_config3.json 2> $O/c3.err; cut -c1-260 $O/bench_config3.json
timeout 900 python bench.py --store-radiation-field --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_radiation_field.json 2> $O/rf.err; cut -c1-260 $O/bench_radiation_field.json
timeout 1200 python bench.py --config 4 --steps 2 --warmup 1 --packets 5e7 --no-cpu-baseline > $O/bench_config4.json 2> $O/c4.err; cut -c1-260 $O/bench_config4.json
timeout 1200 python bench.py --config 5 --steps 2 --warmup 1 --packets 5e7 --no-cpu-baseline > $O/bench_config5.json 2> $O/c5.err; cut -c1-260 $O/bench_config5.json
