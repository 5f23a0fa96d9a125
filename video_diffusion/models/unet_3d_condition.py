from fatezero_amd.video_diffusion.models.unet_3d_condition import *  # noqa: F401,F403
from fatezero_amd.video_diffusion.models.unet_3d_condition import UNetPseudo3DConditionModel  # noqa: F401
