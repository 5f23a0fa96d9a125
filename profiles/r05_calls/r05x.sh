timeout 330 python bench.py --steps 5 --warmup 2 > $O/bench_closing.json 2> $O/bench_closing.err; tail -c 200 $O/bench_closing.err
python - <<'PY'
import json,os
d=json.loads(open(os.environ["O"]+"/bench_closing.json").read().strip().splitlines()[-1])
print("bench:", d["ms_per_step"], d["value"], d["ms_per_step_spread"], d["roofline"]["frac"], d.get("value_normalised"), d["box"].get("flash_calib_hot_us"))
PY
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $O/gpu_tests_closing.log
