export PMC_PROFILE_DUMP=1 PMC_TIMING_DUMP=1
S="PMC_NUM_GROUPS=1,PMC_SERIAL_WALKS=1"
timeout 600 python tools/sweep.py --packets 1e8 default default,$S default 2>&1 | grep -v "PMC_GEN\|amdgpu.ids\|census" | tee gpurun_out/sweep15.txt
PMC_PROP_NO_TRIM=1 timeout 600 python tools/sweep.py --packets 1e8 default default,$S,PMC_WALK_BLOCKS_PER_CU=1 default 2>&1 | grep -v "PMC_GEN\|amdgpu.ids\|census" | tee gpurun_out/sweep15b.txt
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest15.txt 2>&1; grep -E "passed|failed|rror" gpurun_out/pytest15.txt | tail -5
