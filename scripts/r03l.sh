# round-3 GPU call l: BASELINE.md section 3 -- cfg1 (8 f x 256^2 x 10 steps) IN FULL on the CPU oracle of the GPU box's host cores, next to the GPU line
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03l; mkdir -p $O
(timeout 1200 python bench.py --steps 1 --warmup 1 --cpu-cfg1 --no-kernel-breakdown --no-n-edit2-probe) > $O/bench_cfg1.json 2> $O/bench.err
python -c "import json; d=json.load(open('$O/bench_cfg1.json')); print(json.dumps(d['cpu_baseline'], indent=1))"
