#!/bin/bash
# round 3, batch 12 (GPU box): Voronoi with 32-bit cone masks (config 5 at 1e5 and at 1e6 sites), the N = 2 bench path on one
# device (scene file, reduce check; gloo exchange), a one-rank RCCL communicator through bench.py
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03_batch12; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_config5.py -m gpu -x -q > $O/gputests5.txt 2>&1; echo "gpu tests rc=$?"; tail -3 $O/gputests5.txt | cut -c1-200
python bench.py --config 5 --steps 2 --warmup 1 --packets 5e7 --no-cpu-baseline > $O/bench_config5.json 2> $O/bench5.err; echo "config5 rc=$?"; python -c "
import json;d=json.load(open('$O/bench_config5.json'));print(d['value'],d['ms_per_step'],d['roofline']['frac'],d['config']['cells'])"
python bench.py --config 5 --sites 1e6 --steps 2 --warmup 1 --packets 2e7 --no-cpu-baseline > $O/bench_config5_1e6_sites.json 2> $O/bench5m.err; echo "config5 1e6 rc=$?"; tail -2 $O/bench5m.err; python -c "
import json;d=json.load(open('$O/bench_config5_1e6_sites.json'));print(d['value'],d['ms_per_step'],d['roofline']['frac'],d['config']['cells'])"
BENCH_SHARE_DEVICE=1 HSA_ENABLE_IPC_MODE_LEGACY=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 --packets 2e7 > $O/bench_2ranks_one_device.json 2> $O/bench2.err; echo "2 ranks rc=$?"; tail -3 $O/bench2.err; python -c "
import json;d=json.load(open('$O/bench_2ranks_one_device.json'));print(d['value'],d['n_gpus'],d.get('reduce'),d.get('nccl_ranks'),d.get('reduce_check'))"
BENCH_FORCE_COMM=1 python bench.py --steps 2 --warmup 1 --packets 2e7 --no-cpu-baseline --no-secondary > $O/bench_forced_comm.json 2> $O/benchf.err; echo "forced comm rc=$?"; python -c "
import json;d=json.load(open('$O/bench_forced_comm.json'));print(d['value'],d.get('reduce'),d.get('nccl_ranks'),d.get('reduce_check'))"
