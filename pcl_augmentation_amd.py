"""Importable alias of the ``pcl-augmentation_amd`` package (its directory name has a hyphen)."""
import importlib
import sys

_pkg = importlib.import_module("pcl-augmentation_amd")
sys.modules[__name__] = _pkg
