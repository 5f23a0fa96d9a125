set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05a; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests/test_pipeline_gpu.py -x -q -s -k "oracle_executed or whole_job" > $O/geometry.log 2>&1; echo "rc=$?" >> $O/geometry.log
timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "ragged or rejects" > $O/kern.log 2>&1; echo "rc=$?" >> $O/kern.log
timeout 600 python bench.py --steps 3 --warmup 1 > $O/bench.json 2> $O/bench.err; echo "rc=$?" >> $O/bench.err
tail -3 $O/geometry.log; tail -2 $O/kern.log; tail -c 600 $O/bench.json
