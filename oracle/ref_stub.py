"""TEST INFRASTRUCTURE ONLY (build container).

The reference's ``src/models/resnet.py:17-21,976`` imports six torchvision *metadata*
symbols (weight enums, category names, API-usage logger).  torchvision is not installed in
this image and none of those symbols carries arithmetic, so this module registers inert
stand-ins in ``sys.modules`` so that the reference's own Python can be imported on CPU to
generate golden vectors (``gen_golden.py``).  This is our code, not the reference's.
"""
import enum
import sys
import types


def _mod(name):
    m = types.ModuleType(name)
    sys.modules[name] = m
    return m


def install():
    if "torchvision" in sys.modules and not getattr(sys.modules["torchvision"], "_sm3_stub", False):
        return  # a real torchvision exists: nothing to do
    tv = _mod("torchvision")
    tv.__path__ = []
    tv._sm3_stub = True
    tr = _mod("torchvision.transforms")
    tr.__path__ = []
    presets = _mod("torchvision.transforms._presets")
    presets.ImageClassification = type(
        "ImageClassification", (), {"__init__": lambda self, *a, **k: None}
    )
    _mod("torchvision.utils")._log_api_usage_once = lambda obj: None
    models = _mod("torchvision.models")
    models.__path__ = []
    api = _mod("torchvision.models._api")

    class Weights:
        def __init__(self, url=None, transforms=None, meta=None):
            self.url, self.transforms, self.meta = url, transforms, meta

    class WeightsEnum(enum.Enum):
        @classmethod
        def verify(cls, obj):
            if obj is None:
                return None
            if isinstance(obj, str):
                return cls[obj.replace(cls.__name__ + ".", "")]
            return obj

        @property
        def url(self):
            return self.value.url

        @property
        def meta(self):
            return self.value.meta

        def get_state_dict(self, progress=True):
            raise RuntimeError("no network in this container")

    api.Weights, api.WeightsEnum = Weights, WeightsEnum
    _mod("torchvision.models._meta")._IMAGENET_CATEGORIES = [str(i) for i in range(1000)]
    utils = _mod("torchvision.models._utils")

    def _ovewrite_named_param(kwargs, param, new_value):
        if param in kwargs and kwargs[param] != new_value:
            raise ValueError(param)
        kwargs[param] = new_value

    utils._ovewrite_named_param = _ovewrite_named_param
    utils.handle_legacy_interface = lambda **weights: (lambda fn: fn)
    utils._ModelURLs = type("_ModelURLs", (dict,), {})
    if "timm" not in sys.modules:  # src/models/baseline.py:4 imports timm for an arch family the path never uses
        _mod("timm").create_model = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("timm is not installed"))
