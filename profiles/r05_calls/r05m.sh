(timeout 600 python -m pytest tests/test_kernels_gpu.py -q -s -k "layernorm_out") > $O/k.log 2>&1; echo "rc=$?" >> $O/k.log; tail -12 $O/k.log | cut -c1-200
(timeout 900 python -m pytest tests/test_pipeline_gpu.py -q -s -k "fullwidth_sd15 or (whole_job and cfg2)") > $O/p.log 2>&1; echo "rc=$?" >> $O/p.log; grep -E "passed|failed|rc=" $O/p.log | tail -3
timeout 600 python scripts/ab_bench.py fatezero_amd.video_diffusion.models.attention LN_FROM_PRODUCER > $O/ab.txt 2>&1; tail -3 $O/ab.txt
