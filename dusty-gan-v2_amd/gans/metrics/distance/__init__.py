from .cd.chamfer_distance import chamfer_distance  # noqa: F401
from .dcd import calc_dcd as density_aware_chamfer_distance  # noqa: F401
from .emd.earth_mover_distance import EarthMoverDistance, earth_mover_distance  # noqa: F401
