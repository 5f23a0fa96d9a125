"""diffusers/utils (0.11.1) — the few names the reference imports."""
import logging as _pylogging
from collections import OrderedDict
from dataclasses import fields

from . import import_utils
from .import_utils import is_xformers_available


class _Logging:
    @staticmethod
    def get_logger(name):
        return _pylogging.getLogger(name)


logging = _Logging()


def deprecate(*args, **kwargs):
    return None


def is_accelerate_available():
    return False


class BaseOutput(OrderedDict):
    """Dataclass-backed ordered dict with attribute *and* key access (0.11.1 semantics)."""

    def __post_init__(self):
        for f in fields(self):
            v = getattr(self, f.name)
            if v is not None:
                self[f.name] = v

    def __getitem__(self, k):
        if isinstance(k, str):
            return dict(self.items())[k]
        return self.to_tuple()[k]

    def to_tuple(self):
        return tuple(self[k] for k in self.keys())
