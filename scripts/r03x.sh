# round-3 GPU call x: register-resident temporal attention for 24 / 32-frame clips: parity + the cfg5-shaped and 24-frame jobs again
O=gpurun_out/r03x; mkdir -p $O
(timeout 300 python -m pytest tests/test_kernels_gpu.py -q -x -k "temporal") > $O/tests.log 2>&1; tail -2 $O/tests.log
X="--warmup 1 --steps 1 --no-cpu-baseline --no-kernel-breakdown --no-n-edit2-probe"
(timeout 300 python bench.py --frames 32 --latent-size 72 $X) > $O/bench_cfg5_shape.json 2>/dev/null
(timeout 300 python bench.py --frames 24 $X) > $O/bench_24f.json 2>/dev/null
for f in 24f cfg5_shape; do python -c "import json; d=json.load(open('$O/bench_$f.json')); print('$f', round(d['ms_per_step']), 'ms', round(d['value'],3), 'frames/s', d['config']['outputs_finite'])"; done
