from fatezero_amd.video_diffusion.models.clip_text import *  # noqa: F401,F403
