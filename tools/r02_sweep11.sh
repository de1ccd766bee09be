export PMC_PEEL_V1=1
V="default,PMC_WALK_OVERLAP=1"
for g in 2 3; do for w in 1 2; do V="$V default,PMC_NUM_GROUPS=$g,PMC_WALK_BLOCKS_PER_CU=$w"; done; done
V="$V default,PMC_NUM_GROUPS=2,PMC_WALK_BLOCKS_PER_CU=3 default,PMC_NUM_GROUPS=4,PMC_WALK_BLOCKS_PER_CU=1 default,PMC_NUM_GROUPS=4,PMC_WALK_BLOCKS_PER_CU=2 default,PMC_WALK_OVERLAP=1"
timeout 800 python tools/sweep.py --packets 1e8 $V 2>&1 | grep -v "PMC_GEN\|amdgpu.ids" | tee gpurun_out/sweep11.txt
