"""visionworkbench_amd — MI355X-native engine for Vision Workbench's dense block-matching stereo hot path.

The product is libvwgpu.so (hand-written HIP for gfx950 behind the C ABI of include/vwgpu.h); this package is
the thin host-side mirror of the reference's `vw::stereo` entry points used by the tests and the benchmark.
Importing the package does not need a GPU; calling into it does (there is no CPU fallback).
"""
from . import core, synth  # noqa: F401
from .core import (ABSOLUTE_DIFFERENCE, CROSS_CORRELATION, SQUARED_DIFFERENCE, ArgumentErr, BBox2i, Context,  # noqa: F401
                   CostFunctionType, LogicErr, NoImplErr, bounding_box)

__all__ = ["core", "synth", "stereo", "Context", "BBox2i", "CostFunctionType", "bounding_box"]


def __getattr__(name):
    if name == "stereo":
        import importlib
        return importlib.import_module(".stereo", __name__)
    raise AttributeError(name)
