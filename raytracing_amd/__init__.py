"""raytracing_amd: the launchRays() path of the walker-a/RayTracing program on AMD Instinct MI355X.

The package is a thin host-side mirror of the reference's C interface (tracer.py) over libgrafthip.so, the
C-ABI library that holds the hand-written HIP kernels (csrc/grafthip.hip, include/grafthip.h).
"""
from . import _lib  # noqa: F401  (raises if the HIP library has not been built)
from .tracer import (ANTIALIASING_OFF, ANTIALIASING_ON, ORTHOGRAPHIC, PERSPECTIVE, Job, Light, RayTracer, Shape,
                     device_count, frame_pieces, frame_plan, get_diagnostic, multi_stats, init, init_devices, lightInit, lightSetAvoid, planeInit, planeMalloc, sphereInit,
                     sphereMalloc, set_diagnostic, write_png, write_ppm)
from ._lib import GRAFT_OUT_F32_RGBA, GRAFT_OUT_F64_RGB, GRAFT_OUT_U8_RGBA, GraftError

__all__ = ["RayTracer", "Shape", "Light", "Job", "init", "device_count", "sphereMalloc", "sphereInit",
           "planeMalloc", "planeInit", "lightInit", "lightSetAvoid", "ORTHOGRAPHIC", "PERSPECTIVE",
           "ANTIALIASING_OFF", "ANTIALIASING_ON", "GRAFT_OUT_F64_RGB", "GRAFT_OUT_F32_RGBA", "GRAFT_OUT_U8_RGBA",
           "GraftError", "write_ppm", "write_png", "set_diagnostic", "get_diagnostic", "init_devices", "frame_pieces", "frame_plan", "multi_stats"]
