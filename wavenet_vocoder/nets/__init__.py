from .wavenet import *  # noqa: F401,F403  (same re-export as the reference's nets/__init__.py:1)
