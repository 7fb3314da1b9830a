"""Real3D-Aug occlusion handling + insertion merge on MI355X (gfx950).

A drop-in for the per-frame hot path of ctu-vras/pcl-augmentation: the functions of
``Real3DAug/insertion.py`` and ``Real3DAug/tools/closing.py`` keep their names, argument meaning
and error behaviour (``Real3DAug`` sub-package), and ``SceneBatch`` runs many independent scenes
through K inserts with everything resident in HBM.  All compute goes through the C ABI of
``libreal3daug_hip.so`` (``include/real3daug_hip.h``); there is no CPU fallback.

The directory name contains a hyphen, so import it with
``importlib.import_module("pcl-augmentation_amd")`` (or through ``pcl_augmentation_amd.py`` at
the repository root).
"""
from . import _lib, affinity, synth  # noqa: F401
from ._lib import R3DError  # noqa: F401
from .batch import SceneBatch, augment_batch, run_sharded, shard_indices  # noqa: F401
from . import Real3DAug  # noqa: F401
from .pipeline import AugmentPipeline, Frame, run_sharded_files  # noqa: F401
from . import places  # noqa: F401
from .places import PlaceScene, find_places  # noqa: F401
from .placed import PlacedInserter  # noqa: F401
from .rich_map import build_rich_map  # noqa: F401

__all__ = ["SceneBatch", "augment_batch", "run_sharded", "shard_indices", "AugmentPipeline", "Frame", "run_sharded_files", "Real3DAug", "synth", "R3DError", "places", "PlaceScene", "find_places", "PlacedInserter", "build_rich_map", "affinity"]
