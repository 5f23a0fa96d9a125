FZ_FULL_PARITY=1 FZ_PARITY_DUMP=$O timeout 3000 python -m pytest tests/test_pipeline_gpu.py -x -q -s -k "judged_job" > $O/deep_parity.log 2>&1; tail -30 $O/deep_parity.log | cut -c1-600
