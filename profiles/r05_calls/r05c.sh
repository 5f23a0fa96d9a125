cd $GRAFT_REPO_ROOT
O=gpurun_out/r05c; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1800 python -m pytest tests/test_pipeline_gpu.py -q -s -k "whole_job and (cfg4 or cfg5 or cfg2)" > $O/geometry.log 2>&1; echo "rc=$?" >> $O/geometry.log
timeout 600 python bench.py --steps 3 --warmup 1 > $O/bench.json 2> $O/bench.err; echo "rc=$?" >> $O/bench.err
grep -E "passed|failed|rc=" $O/geometry.log | tail -5; tail -c 300 $O/bench.json; tail -3 $O/bench.err
