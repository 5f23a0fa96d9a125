# round-3 GPU call u: kernel stats of the cfg5-shaped job (32 f x 576^2) -- why is it slower than in round 2?
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03u; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python $R/bench.py --frames 32 --latent-size 72 --ddim-steps 10 --warmup 0 --steps 1 --no-cpu-baseline --no-kernel-breakdown --no-n-edit2-probe > $O/bench.json 2> $O/bench.err
cd $R
f=$(ls $O/prof/*/bench_kernel_stats.csv $O/prof/bench_kernel_stats.csv 2>/dev/null | head -1); cp "$f" $O/kernel_stats.csv; rm -rf $O/prof
head -c 300 $O/bench.json; echo; head -14 $O/kernel_stats.csv | cut -c1-170
