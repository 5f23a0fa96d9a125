"""Install empty stand-ins for viz/IO-only third-party modules the reference imports at module scope."""
import sys, types


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install():
    try:  # must be imported before torchvision is stubbed (it probes find_spec("torchvision"))
        import transformers  # noqa: F401
        from transformers import CLIPTextModel, CLIPTokenizer  # noqa: F401
    except Exception:
        pass
    if "cv2" not in sys.modules:
        _mod("cv2", FONT_HERSHEY_SIMPLEX=0)
    if "omegaconf" not in sys.modules:
        class DictConfig(dict):
            pass
        dc = _mod("omegaconf.dictconfig", DictConfig=DictConfig)
        _mod("omegaconf", dictconfig=dc, DictConfig=DictConfig, OmegaConf=object)
    if "torchvision" not in sys.modules:
        def _save_image(*a, **k):
            raise RuntimeError("torchvision stub: save_image must not be reached (set save_path=None)")
        u = _mod("torchvision.utils", save_image=_save_image)
        t = _mod("torchvision.transforms")
        _mod("torchvision", utils=u, transforms=t)
    if "imageio" not in sys.modules:
        _mod("imageio")
    if "ftfy" not in sys.modules:
        _mod("ftfy", fix_text=lambda s: s)
    if "requests" not in sys.modules:
        _mod("requests")
