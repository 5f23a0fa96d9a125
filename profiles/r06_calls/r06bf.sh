python scripts/ab_bench.py fatezero_amd.video_diffusion.models.resnet CONV_UP2 > $O/conv_up2_job_ab.txt 2>&1; tail -3 $O/conv_up2_job_ab.txt
timeout 1500 python -m pytest tests -x -q -m gpu -k "pipeline or fullsize or parity or unet" 2>&1 | tail -3
