timeout 600 python -m pytest tests/test_pipeline_gpu.py -x -q -s -k "disk_store" 2>&1 | tail -8
timeout 900 python -m pytest tests/test_pipeline_gpu.py -x -q -s -k "cfg2_fullwidth_8f_all_stored" 2>&1 | tail -12
