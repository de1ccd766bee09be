python - <<'P'
t=open('tests/ski/cfg2.ski').read().replace('storeRadiationField="false"','storeRadiationField="true"')
open('/tmp/cfg2.ski','w').write(t)
P
timeout 600 python tools/sweep.py --ski /tmp/cfg2.ski --packets 1e8 default libpmc_norfatomic.so default libpmc_norfatomic.so 2>&1 | grep pkt | tee gpurun_out/sweep24.txt
