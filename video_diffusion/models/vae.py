from fatezero_amd.video_diffusion.models.vae import *  # noqa: F401,F403
