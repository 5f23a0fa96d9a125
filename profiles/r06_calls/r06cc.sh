R=$GRAFT_REPO_ROOT; A=$R/$O
for cfg in "16 64" "24 64" "32 72"; do set -- $cfg; (timeout 500 python bench.py --steps 1 --warmup 1 --frames $1 --latent-size $2 --no-cpu-baseline --no-kernel-breakdown --no-n-edit2-probe --no-split-mask) > $A/bench_$1f_$2.json 2> $A/bench_$1f_$2.err; python -c "
import json;d=json.loads(open('$A/bench_$1f_$2.json').read().strip().splitlines()[-1]);print('$1 f x $2: ', d['ms_per_step'], d['value'], d['box']['flash_calib_hot_us'], d['box']['during_timed_region']['sclk_MHz'])"; done
