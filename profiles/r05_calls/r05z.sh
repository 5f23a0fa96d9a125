timeout 200 python -m pytest tests/test_pipeline_gpu.py -x -q -k "issue_plans or disk_store" 2>&1 | tail -4
