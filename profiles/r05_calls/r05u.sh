FZ_ISSUE_PLANS=1 timeout 400 python scripts/ab_bench.py fatezero_amd.issue ENABLED 2>&1 | tail -2 | tee -a $O/ab.txt
FZ_ISSUE_PLANS=1 FZ_ISSUE_RUNAHEAD=3 timeout 400 python scripts/ab_bench.py fatezero_amd.issue ENABLED 2>&1 | tail -2 | tee -a $O/ab.txt
FZ_ISSUE_PLANS=1 FZ_ISSUE_GRAPH=0 timeout 400 python scripts/ab_bench.py fatezero_amd.issue ENABLED 2>&1 | tail -2 | tee -a $O/ab.txt
