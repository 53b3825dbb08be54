from .furthest_point_sampling import *  # noqa: F401,F403
