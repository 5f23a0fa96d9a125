from fatezero_amd.video_diffusion.models import *  # noqa: F401,F403
