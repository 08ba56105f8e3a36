from .misc import *  # noqa: F401,F403
