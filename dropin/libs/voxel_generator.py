"""`from libs.voxel_generator import Voxelization` (libs/dataset.py:6)."""
from pcaccumulation_amd.voxel_generator import Voxelization  # noqa: F401
