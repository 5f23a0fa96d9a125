from fatezero_amd.video_diffusion.models.attention import *  # noqa: F401,F403
