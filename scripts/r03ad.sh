# round-3 GPU call ad: the two latent-blend tests again (applied-mask aware gate)
O=gpurun_out/r03ad; mkdir -p $O
(timeout 300 python -m pytest tests/test_pipeline_gpu.py -q -s -k "latentblend") > $O/tests.log 2>&1
grep -n "pipe_refine_reweight_latentblend {" $O/tests.log | cut -c1-1200 | head -1
tail -4 $O/tests.log
