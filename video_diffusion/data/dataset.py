from fatezero_amd.video_diffusion.data.dataset import *  # noqa: F401,F403
