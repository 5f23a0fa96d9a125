(timeout 900 python -m pytest tests/test_pipeline_gpu.py -q -s -k "(whole_job and (cfg4 or cfg2)) or long_clips_vs" --durations=8) > $O/t.log 2>&1; echo "rc=$?" >> $O/t.log
grep -E "passed|failed|rc=|s call" $O/t.log | tail -12; grep -o "'applied_mask_flips_vs_fp32[a-z_]*': [0-9]*\|'applied_mask_step_total': [0-9]*\|fullwidth forward.*" $O/t.log
