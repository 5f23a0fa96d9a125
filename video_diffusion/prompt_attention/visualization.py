from fatezero_amd.video_diffusion.prompt_attention.visualization import *  # noqa: F401,F403
