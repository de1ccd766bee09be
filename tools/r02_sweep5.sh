timeout 800 python tools/sweep.py --packets 1e8 default default default,PMC_NUM_GROUPS=2 default,PMC_NUM_GROUPS=1 default,PMC_LAUNCH_BLOCKS_PER_CU=2 default,PMC_LAUNCH_BLOCKS_PER_CU=8 2>&1 | grep -v "PMC_GEN\|amdgpu.ids" | tee gpurun_out/sweep5.txt
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee gpurun_out/pytest5.txt
