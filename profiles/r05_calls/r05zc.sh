timeout 80 python -m pytest tests/test_pipeline_gpu.py -x -q -k "issue_plans" 2>&1 | tail -3
