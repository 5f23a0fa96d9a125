# round-3 GPU call m: the frame-sharded pipeline on the real kernels (two gloo ranks sharing the GPU) + the GPU number of BASELINE cfg1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03m; mkdir -p $O
(timeout 600 python -m pytest tests/test_dist_gpu.py -q -x -s) > $O/dist_gpu.log 2>&1; tail -6 $O/dist_gpu.log | cut -c1-600
(timeout 300 python bench.py --cfg1 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-breakdown) > $O/bench_cfg1_gpu.json 2> $O/bench.err; head -c 700 $O/bench_cfg1_gpu.json; echo; tail -3 $O/bench.err
