# round-3 GPU call p: GPU suite (without the two 5-minute full-width oracle cases) on the final library
O=gpurun_out/r03p; mkdir -p $O
(timeout 900 python -m pytest tests -q -x -m gpu -k "not fullwidth") > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log
(timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-breakdown --no-n-edit2-probe) > $O/bench.json 2>/dev/null; head -c 330 $O/bench.json; echo
