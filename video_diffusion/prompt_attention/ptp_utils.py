from fatezero_amd.video_diffusion.prompt_attention.ptp_utils import *  # noqa: F401,F403
