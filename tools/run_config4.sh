export TMPDIR=/tmp
O=$PWD/gpurun_out/r01_c4; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -5 $O/pytest.log
nproc
timeout 900 python bench.py --config 4 --steps 2 --warmup 1 --packets 5e7 > $O/bench_c4.json 2> $O/bench_c4.err; cat $O/bench_c4.json; tail -3 $O/bench_c4.err
timeout 600 python bench.py --source uniform --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_uniform.json 2> $O/bench_uniform.err; cat $O/bench_uniform.json
