timeout 600 python -m pytest tests/test_pipeline_gpu.py -x -q -s -k "disk_store" 2>&1 | tail -15
