timeout 300 python tools/sweep.py --packets 1e8 default default default > gpurun_out/sweep18.txt 2>&1
PMC_PROP_NO_TRIM=1 timeout 300 python tools/sweep.py --packets 1e8 default default default >> gpurun_out/sweep18.txt 2>&1
grep pkt gpurun_out/sweep18.txt
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest18.txt 2>&1; grep -E "passed|failed|rror" gpurun_out/pytest18.txt | tail -5
