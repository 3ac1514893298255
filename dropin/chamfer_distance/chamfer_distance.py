"""`from chamfer_distance.chamfer_distance import ChamferDistance` (models/tpointnet.py:5) without the CUDA JIT build."""
from pcaccumulation_amd.chamfer_distance import ChamferDistance, ChamferDistanceFunction  # noqa: F401
