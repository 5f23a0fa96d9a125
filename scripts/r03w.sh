# round-3 GPU call w: cfg5-shaped job (32 f x 576^2 x 50+50 steps) after ONE full warm-up job (the 30-224 GB map arena is allocated there and recycled by the timed job), and the 16 / 24-frame jobs the same way
O=gpurun_out/r03w; mkdir -p $O
X="--warmup 1 --steps 1 --no-cpu-baseline --no-kernel-breakdown --no-n-edit2-probe"
(timeout 300 python bench.py --frames 32 --latent-size 72 $X) > $O/bench_cfg5_shape.json 2>/dev/null
(timeout 200 python bench.py --frames 16 $X) > $O/bench_16f.json 2>/dev/null
(timeout 300 python bench.py --frames 24 $X) > $O/bench_24f.json 2>/dev/null
for f in 16f 24f cfg5_shape; do python -c "import json; d=json.load(open('$O/bench_$f.json')); print('$f', round(d['ms_per_step']), 'ms', round(d['value'],3), 'frames/s', d['config']['arena_GB'], 'GB', d['config']['outputs_finite'])"; done
