"""diffusers/modeling_utils.py (0.11.1): ModelMixin — only .device/.dtype are used by the reference."""
import torch


class ModelMixin(torch.nn.Module):
    _supports_gradient_checkpointing = False

    @property
    def device(self):
        return next(self.parameters()).device

    @property
    def dtype(self):
        return next(self.parameters()).dtype
