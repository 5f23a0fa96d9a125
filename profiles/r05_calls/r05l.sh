(timeout 1700 python -m pytest tests -m gpu -q -s --durations=12) > $O/gpu_tests.log 2>&1; echo "rc=$?" >> $O/gpu_tests.log
grep -E "passed|failed|rc=" $O/gpu_tests.log | tail -3
timeout 600 python bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err; echo "rc=$?" >> $O/bench.err; tail -2 $O/bench.err; head -c 600 $O/bench.json
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
