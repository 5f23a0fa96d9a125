from fatezero_amd.video_diffusion.models.lora import *  # noqa: F401,F403
