from fatezero_amd.video_diffusion.models.unet_3d_blocks import *  # noqa: F401,F403
