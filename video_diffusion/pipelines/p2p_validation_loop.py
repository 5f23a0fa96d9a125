from fatezero_amd.video_diffusion.pipelines.p2p_validation_loop import *  # noqa: F401,F403
from fatezero_amd.video_diffusion.pipelines.p2p_validation_loop import P2pSampleLogger, tensor_to_numpy  # noqa: F401
