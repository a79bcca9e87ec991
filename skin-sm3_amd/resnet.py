"""Top-level `resnet` module that the reference's inference.py imports (`import resnet`, inference.py:4,36):
a re-export of src/models/resnet.py, where the encoder mirror lives."""
from src.models.resnet import *  # noqa: F401,F403
from src.models.resnet import __all__  # noqa: F401
