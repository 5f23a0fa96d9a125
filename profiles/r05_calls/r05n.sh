(timeout 600 python -m pytest tests/test_kernels_gpu.py -q -s -k "layernorm_out") > $O/k.log 2>&1; echo "rc=$?" >> $O/k.log; grep -E "gemm_lnout|passed|failed" $O/k.log | cut -c1-220
(timeout 900 python -m pytest tests/test_pipeline_gpu.py -q -s -k "fullwidth_sd15") > $O/p.log 2>&1; echo "rc=$?" >> $O/p.log; grep -E "passed|failed|rc=" $O/p.log | tail -3
timeout 600 python scripts/ab_bench.py fatezero_amd.video_diffusion.models.attention LN_FROM_PRODUCER > $O/ab.txt 2>&1; tail -3 $O/ab.txt
