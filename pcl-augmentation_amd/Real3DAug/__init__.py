"""Mirror of the reference's ``Real3DAug`` package for the hot path (same module names)."""
from . import insertion  # noqa: F401
from .tools import closing, cut_bbox, datasets, find_spot, find_spot_od  # noqa: F401
