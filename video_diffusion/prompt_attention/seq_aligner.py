from fatezero_amd.video_diffusion.prompt_attention.seq_aligner import *  # noqa: F401,F403
