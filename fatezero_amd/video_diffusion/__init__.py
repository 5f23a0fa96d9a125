"""MI355X-native mirror of the reference's `video_diffusion` package for the DDIM-inversion -> denoise path.

Same import paths, class names, constructor signatures and state_dict key names as ChenyangQiQi/FateZero
(`/root/reference/video_diffusion`), different internals: activations are token-major fp16, every hot op goes
through the C ABI of libfatezero_hip.so (fatezero_amd/kernels.py), attention maps live in an HBM arena and the
controllers are descriptors consumed by fused kernels.  The top-level `video_diffusion` package of this repo
re-exports this one so that the reference's YAML `target:` strings resolve unchanged.
"""
