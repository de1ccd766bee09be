PMC_GEN_DUMP=1 timeout 300 python tools/sweep.py --packets 1e8 default default default > gpurun_out/sweep20.txt 2>&1
grep pkt gpurun_out/sweep20.txt
grep PMC_GEN gpurun_out/sweep20.txt | tail -60 | awk '{printf "%s g%s live=%s w=%s t=%s | ", $2, $4, $6, $8, $10; if (NR%3==0) printf "\n"}'
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest20.txt 2>&1; grep -E "passed|failed|rror" gpurun_out/pytest20.txt | tail -5
