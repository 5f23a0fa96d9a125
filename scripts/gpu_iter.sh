# One measurement iteration on the MI355X (run through gpurun from the repo root):  scripts/gpu_iter.sh <tag> [tests-k-expr]
#   GPU parity of the selected kernel cases, the GEMM / conv sweeps, the bench under rocprofv3 (kernel stats) -> gpurun_out/<tag>/
TAG=${1:-iter}; KEXPR=${2:-"gemm or conv"}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
(timeout 300 python -m pytest tests/test_kernels_gpu.py -q -x -k "$KEXPR") > $O/ktests.log 2>&1; tail -3 $O/ktests.log
if [ "$SWEEP" != "0" ]; then
(timeout 250 python scripts/kbench.py --gemm) > $O/kbench_gemm.json 2> $O/kbench_gemm.err
(timeout 300 python scripts/kbench.py --conv) > $O/kbench_conv.json 2> $O/kbench_conv.err
fi
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python $R/bench.py --no-cpu-baseline > $O/bench_prof.json 2> $O/bench_prof.err
cd $R
f=$(ls $O/prof/*/bench_kernel_stats.csv $O/prof/bench_kernel_stats.csv 2>/dev/null | head -1); cp "$f" $O/kernel_stats.csv; rm -rf $O/prof
if [ "$PLAIN" = "1" ]; then (timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline) > $O/bench.json 2> $O/bench.err; fi
head -c 600 $O/bench_prof.json; echo; tail -2 $O/bench_prof.err
