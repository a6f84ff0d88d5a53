from .base import RasterBlock, BaseSingle  # noqa: F401
from .elemwise import *  # noqa: F401,F403
from .misc import *  # noqa: F401,F403
from .sources import *  # noqa: F401,F403
from .parallelize import *  # noqa: F401,F403
from .spatial import *  # noqa: F401,F403
from .reduction import *  # noqa: F401,F403
from .combine import *  # noqa: F401,F403
from .temporal import *  # noqa: F401,F403
