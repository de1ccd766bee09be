#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
python __graft_entry__.py smoke 2>&1 | tail -2
T0=$(date +%s); python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_driver.json 2> gpurun_out/bench_driver.err; echo "Elapsed (wall clock) $(( $(date +%s) - T0 )) s" >> gpurun_out/bench_driver.err
grep "Elapsed (wall clock)" gpurun_out/bench_driver.err
python -c "
import json; d=json.loads([l for l in open('gpurun_out/bench_driver.json') if l.startswith('{')][-1]); r=d['roofline']; print('%.4g pkt/s %.1f ms frac %.3f in-run %s traffic/alg %.3f cpu %s'%(d['value'], d['ms_per_step'], r['frac'], r['traffic_measured_in_run'], r.get('traffic_over_algorithmic', 0), d['cpu_baseline']['value']))"
