"""`from models.cluster import Cluster` (models/motionnet.py:10) resolved to the MI355X implementation."""
from pcaccumulation_amd.cluster import Cluster  # noqa: F401
