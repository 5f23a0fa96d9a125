from fatezero_amd.video_diffusion.data.transform import *  # noqa: F401,F403
