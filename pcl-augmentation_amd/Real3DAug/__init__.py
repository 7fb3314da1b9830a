"""Mirror of the reference's ``Real3DAug`` package for the hot path (same module names)."""
from . import insertion  # noqa: F401
from .tools import closing, cut_bbox, datasets, find_spot  # noqa: F401
from . import tools as _tools
_tools.find_spot_od = find_spot.od          # (the object_detection flavour: one module, `find_spot.od`)
