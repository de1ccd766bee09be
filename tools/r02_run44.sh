export PMC_PROFILE_DUMP=1 PMC_TIMING_DUMP=1
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -m gpu -x -q > gpurun_out/pytest44.txt 2>&1; grep -E "passed|failed|rror|assert" gpurun_out/pytest44.txt | tail -5
S="PMC_NUM_GROUPS=1,PMC_WALK_BLOCKS_PER_CU=3,PMC_SERIAL_WALKS=1"
timeout 700 python tools/sweep.py --packets 1e8 default default libpmc_fullrec.so libpmc_fullrec.so default,$S libpmc_fullrec.so,$S default,$S,PMC_PEEL_BLOCKS_PER_CU=2 2>&1 | grep "pkt\|TIMING peel 1\|TIMING peel 2" | tee gpurun_out/sweep44.txt
