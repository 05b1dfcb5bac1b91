from .utils import *  # noqa  (same re-export as the reference's wavenet_vocoder/utils/__init__.py)
