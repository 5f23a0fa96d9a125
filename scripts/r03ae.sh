# round-3 GPU call ae: the GPU test modules the closing run had not covered (full-size properties, frame-sharded ranks on one GPU, VAE, CLIP, CLI)
O=gpurun_out/r03ae; mkdir -p $O
(timeout 420 python -m pytest tests/test_fullsize_properties_gpu.py tests/test_dist_gpu.py tests/test_vae_gpu.py tests/test_clip_gpu.py tests/test_clip_text_gpu.py tests/test_cli_gpu.py -q --durations=10) > $O/tests.log 2>&1
tail -18 $O/tests.log
