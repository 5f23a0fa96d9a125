R=$GRAFT_REPO_ROOT; A=$R/$O
( time timeout 1500 python -m pytest tests/ -x -q -m gpu ) > $A/gpu_tests_full.log 2>&1; tail -6 $A/gpu_tests_full.log
(timeout 600 python bench.py) > $A/bench_default.json 2> $A/bench_default.err; head -c 300 $A/bench_default.json; echo
