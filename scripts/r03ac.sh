# round-3 GPU call ac: the pipeline tests (incl. both full-width oracle comparisons) after the flash / GroupNorm changes, with the
# latent-blend max error split into flipped / unflipped mask pixels
O=gpurun_out/r03ac; mkdir -p $O
(timeout 700 python -m pytest tests/test_pipeline_gpu.py -q -s --durations=8) > $O/pipeline_tests.log 2>&1
grep -n "pipe_refine_reweight_latentblend {" $O/pipeline_tests.log | cut -c1-900
tail -14 $O/pipeline_tests.log
