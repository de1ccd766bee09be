export PMC_PEEL_V1=1
timeout 600 python tools/sweep.py --packets 1e8 default libpmc_k8.so libpmc_k4.so default libpmc_k8.so,PMC_NUM_GROUPS=2 libpmc_k8.so,PMC_NUM_GROUPS=2,PMC_WALK_BLOCKS_PER_CU=2 2>&1 | grep -v "PMC_GEN\|amdgpu.ids\|census" | tee gpurun_out/sweep16.txt
unset PMC_PEEL_V1
timeout 600 python tools/sweep.py --packets 1e8 libpmc_k8.so libpmc_k4.so 2>&1 | grep -v "PMC_GEN\|amdgpu.ids\|census" | tee gpurun_out/sweep16b.txt
