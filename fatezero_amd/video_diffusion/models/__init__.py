from .unet_3d_condition import UNetPseudo3DConditionModel  # noqa: F401
