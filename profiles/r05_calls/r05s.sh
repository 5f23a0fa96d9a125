timeout 900 python -m pytest tests/test_pipeline_gpu.py -x -q -s -k "issue_plans" 2>&1 | tail -8
for c in none store; do timeout 300 python scripts/host_issue_time.py --frames 8 --controller $c 2>&1 | tail -1 | tee -a $O/host_issue_time_graph.txt; done
timeout 300 python scripts/host_issue_time.py --frames 1 --controller store 2>&1 | tail -1 | tee -a $O/host_issue_time_graph.txt
FZ_ISSUE_GRAPH=0 timeout 300 python scripts/host_issue_time.py --frames 1 --controller store 2>&1 | tail -1 | tee -a $O/host_issue_time_graph.txt
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-breakdown --no-n-edit2-probe --no-box --no-split-mask"
timeout 300 $B --issue-plans > $O/bench_plans.json 2> $O/bench_plans.err; tail -c 300 $O/bench_plans.err
timeout 300 $B > $O/bench_walk.json 2> $O/bench_walk.err; tail -c 300 $O/bench_walk.err
python - <<'PY'
import json,os
O=os.environ["O"]
for n in ("walk","plans"):
    try:
        d=json.loads(open(f"{O}/bench_{n}.json").read().strip().splitlines()[-1])
        print(n, d["ms_per_step"], d["ms_per_step_spread"], d.get("issue_plans"), d["roofline"]["frac"], d["roofline"].get("launches"), d["config"]["outputs_finite"])
    except Exception as e: print(n, "failed", e)
PY
