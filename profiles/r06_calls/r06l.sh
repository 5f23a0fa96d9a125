python scripts/ab_matrix.py fatezero_amd.video_diffusion.models.attention:FF_CHAIN --rounds=4 > $O/ff_chain_job_ab.txt 2>&1; tail -3 $O/ff_chain_job_ab.txt
