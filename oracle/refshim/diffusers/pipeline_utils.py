"""diffusers/pipeline_utils.py (0.11.1): the slice of DiffusionPipeline the reference pipelines touch."""
import torch
from tqdm.auto import tqdm

from .configuration_utils import ConfigMixin


class DiffusionPipeline(ConfigMixin):
    def register_modules(self, **kwargs):
        for name, module in kwargs.items():
            setattr(self, name, module)
        self.register_to_config(**{k: (type(v).__module__, type(v).__name__) for k, v in kwargs.items()})

    @property
    def _execution_device(self):
        return next(self.unet.parameters()).device

    @property
    def device(self):
        return next(self.unet.parameters()).device

    def progress_bar(self, iterable=None, total=None):
        cfg = getattr(self, "_progress_bar_config", {})
        if iterable is not None:
            return tqdm(iterable, **cfg)
        return tqdm(total=total, **cfg)

    def set_progress_bar_config(self, **kwargs):
        self._progress_bar_config = kwargs

    @staticmethod
    def numpy_to_pil(images):
        return images
