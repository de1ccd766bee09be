timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py -m gpu -x -q -k "radiation or rf or multi or reduce" > gpurun_out/pytest25.txt 2>&1; grep -E "passed|failed|rror|assert" gpurun_out/pytest25.txt | tail -8
python - <<'P'
t=open('tests/ski/cfg2.ski').read().replace('storeRadiationField="false"','storeRadiationField="true"')
open('/tmp/cfg2.ski','w').write(t)
P
timeout 600 python tools/sweep.py --ski /tmp/cfg2.ski --packets 1e8 default default default,PMC_RF_ATOMICS=1 2>&1 | grep pkt | tee gpurun_out/sweep25.txt
timeout 300 python tools/sweep.py --packets 1e8 default default 2>&1 | grep pkt | tee -a gpurun_out/sweep25.txt
