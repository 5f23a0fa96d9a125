R=$GRAFT_REPO_ROOT; A=$R/$O
timeout 900 python -m pytest tests/test_pipeline_gpu.py -x -q -k "cfg4_attribute_24f" 2>&1 | tail -3
( time timeout 1500 python -m pytest tests/ -x -q -m gpu ) > $A/gpu_tests_full.log 2>&1; tail -6 $A/gpu_tests_full.log
