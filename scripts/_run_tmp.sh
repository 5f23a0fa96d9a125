R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02l; mkdir -p $O
(timeout 600 python -m pytest tests/test_vae_gpu.py tests/test_clip_text_gpu.py tests/test_cli_gpu.py -q -x -s) > $O/rows_f.log 2>&1; tail -12 $O/rows_f.log
