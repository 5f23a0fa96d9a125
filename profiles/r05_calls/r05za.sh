timeout 150 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_final_check.json 2> $O/bench_final_check.err; tail -c 300 $O/bench_final_check.err
python - <<'PY'
import json,os
d=json.loads(open(os.environ["O"]+"/bench_final_check.json").read().strip().splitlines()[-1])
print("bench:", d["ms_per_step"], d["value"], d["roofline"]["frac"], d["box"]["during_timed_region"]["sclk_MHz"], d["split_mask_job"]["stored_rows_fraction"], len(d["rooflines"]), d["config_faithful_n_edit_2"]["value"])
PY
