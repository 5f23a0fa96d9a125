"""diffusers/configuration_utils.py (0.11.1): FrozenDict, ConfigMixin, register_to_config (restated)."""
import functools
import inspect
from collections import OrderedDict


class FrozenDict(OrderedDict):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        for key, value in self.items():
            object.__setattr__(self, key, value)


class ConfigMixin:
    config_name = None

    def register_to_config(self, **kwargs):
        kwargs.pop("kwargs", None)
        if not hasattr(self, "_internal_dict"):
            internal_dict = kwargs
        else:
            internal_dict = {**self._internal_dict, **kwargs}
        self._internal_dict = FrozenDict(internal_dict)

    @property
    def config(self):
        return self._internal_dict


def register_to_config(init):
    @functools.wraps(init)
    def inner_init(self, *args, **kwargs):
        init_kwargs = {k: v for k, v in kwargs.items() if not k.startswith("_")}
        init(self, *args, **init_kwargs)
        signature = inspect.signature(init)
        parameters = {
            name: p.default for i, (name, p) in enumerate(signature.parameters.items())
            if i > 0 and p.kind not in (p.VAR_KEYWORD, p.VAR_POSITIONAL)
        }
        new_kwargs = {}
        for arg, name in zip(args, parameters.keys()):
            new_kwargs[name] = arg
        new_kwargs.update({k: init_kwargs.get(k, default) for k, default in parameters.items() if k not in new_kwargs})
        # extra **kwargs of the wrapped __init__ are recorded too (0.11.1 behaviour)
        new_kwargs.update({k: v for k, v in init_kwargs.items() if k not in new_kwargs})
        getattr(self, "register_to_config")(**new_kwargs)

    return inner_init
