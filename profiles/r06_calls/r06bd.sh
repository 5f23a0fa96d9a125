timeout 1500 python -m pytest tests -x -q -m gpu > $O/gpu_tests_full.log 2>&1; tail -5 $O/gpu_tests_full.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 1500 $O/bench_default.json
