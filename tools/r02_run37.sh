timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/pytest37.txt 2>&1; grep -E "passed|failed|rror" gpurun_out/pytest37.txt | tail -3
timeout 1200 python bench.py --config 5 --steps 2 --warmup 1 --packets 5e7 --no-cpu-baseline > gpurun_out/bench_config5.json 2> gpurun_out/c5.err; cut -c1-250 gpurun_out/bench_config5.json
