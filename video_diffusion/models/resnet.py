from fatezero_amd.video_diffusion.models.resnet import *  # noqa: F401,F403
