# Re-exported operator API, same surface as the reference's gans/models/ops/__init__.py:1-5.
from .common import *  # noqa: F401,F403
from .fourier import *  # noqa: F401,F403
from .fused_act.fused_act import *  # noqa: F401,F403
from .gumbel import *  # noqa: F401,F403
from .style import *  # noqa: F401,F403
