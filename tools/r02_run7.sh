timeout 600 python tools/sweep.py --packets 1e8 default default default,PMC_NUM_GROUPS=1,PMC_WALK_BLOCKS_PER_CU=3 2>&1 | grep -v "PMC_GEN\|amdgpu.ids" | tee gpurun_out/sweep7.txt
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest7.txt 2>&1; grep -E "passed|failed|rror" gpurun_out/pytest7.txt | tail -5
