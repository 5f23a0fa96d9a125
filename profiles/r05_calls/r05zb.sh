timeout 100 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 100 python -m pytest tests/test_pipeline_gpu.py -x -q -k "issue_plans" 2>&1 | tail -2
