export TMPDIR=/tmp
O=$PWD/gpurun_out/r01_c5; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -5 $O/pytest.log
timeout 900 python bench.py --config 5 --steps 2 --warmup 1 --packets 5e7 > $O/bench_c5.json 2> $O/bench_c5.err; cat $O/bench_c5.json; tail -3 $O/bench_c5.err
