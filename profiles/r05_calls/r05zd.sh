FZ_ISSUE_PLANS=1 timeout 65 python scripts/ab_bench.py fatezero_amd.issue ENABLED 2>&1 | tail -3 | tee $O/ab_final.txt
