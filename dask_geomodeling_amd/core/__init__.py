from .graphs import *  # noqa: F401,F403
