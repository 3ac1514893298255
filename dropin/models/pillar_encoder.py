"""models/pillar_encoder.py import path (models/motionnet.py:3, models/stpn.py:4)."""
from pcaccumulation_amd.pillar_encoder import (PillarFeatureNet, ResnetBlockFC, scatter_point_pillar,  # noqa: F401
                                               inverse_scatter_point_pillar, temporal_ungrid, ungrid)
