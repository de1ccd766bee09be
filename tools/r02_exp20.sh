#!/bin/bash
# round 2, experiment 20: product multi-device path (RCCL), overflow error, bench.py with the secondary workload
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r02_exp20; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_multi.py -m gpu -x -q > $O/pytest_multi.log 2>&1; tail -15 $O/pytest_multi.log
timeout 600 python bench.py --steps 3 --warmup 1 > $O/bench.json 2> $O/bench.err; cat $O/bench.json; tail -3 $O/bench.err
