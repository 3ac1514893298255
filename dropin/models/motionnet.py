"""`from models.motionnet import MotionNet` (main.py:8) resolved to the MI355X implementation."""
from pcaccumulation_amd.motionnet import MotionNet, MIN_POINTS  # noqa: F401
