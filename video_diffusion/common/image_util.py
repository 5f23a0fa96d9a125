from fatezero_amd.video_diffusion.common.image_util import *  # noqa: F401,F403
