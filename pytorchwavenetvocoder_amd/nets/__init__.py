from .wavenet import *  # noqa  (same re-export as the reference's wavenet_vocoder/nets/__init__.py:1)
