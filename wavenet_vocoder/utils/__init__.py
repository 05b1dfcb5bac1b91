from .utils import *  # noqa: F401,F403  (same re-export as the reference's utils/__init__.py)
