# round-3 GPU call t: the larger BASELINE shapes as complete jobs on ONE GPU (no warm-up job), as in round 2
O=gpurun_out/r03t; mkdir -p $O
X="--warmup 0 --steps 1 --no-cpu-baseline --no-kernel-breakdown --no-n-edit2-probe"
(timeout 200 python bench.py --frames 16 $X) > $O/bench_16f.json 2>/dev/null
(timeout 300 python bench.py --frames 24 $X) > $O/bench_24f.json 2>/dev/null
(timeout 300 python bench.py --frames 32 --latent-size 72 $X) > $O/bench_cfg5_shape.json 2>/dev/null
for f in 16f 24f cfg5_shape; do python -c "import json; d=json.load(open('$O/bench_$f.json')); print('$f', round(d['ms_per_step']), 'ms', round(d['value'],3), 'frames/s', d['config']['arena_GB'], 'GB', d['config']['outputs_finite'])"; done
