O=$PWD/gpurun_out/${RUN_NAME:-r01_v8b}_other; rm -rf $O; mkdir -p $O
timeout 900 python bench.py --config 4 --steps 2 --warmup 1 --packets 5e7 > $O/bench_config4.json 2> $O/c4.err
timeout 900 python bench.py --config 5 --steps 2 --warmup 1 --packets 5e7 > $O/bench_config5.json 2> $O/c5.err
timeout 600 python bench.py --source uniform --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_uniform_source.json 2> $O/u.err
for f in $O/*.json; do python -c "
import json,sys
d=json.load(open('$f')); r=d['roofline']; print('$f'.split('/')[-1], '%.4g'%d['value'], 'frac %.4f'%r['frac'], (d.get('cpu_baseline') or {}).get('value'))"; done
