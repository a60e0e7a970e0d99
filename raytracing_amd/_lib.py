"""ctypes view of libgrafthip.so (the C ABI declared in include/grafthip.h).

There is no CPU fallback: if the shared library is missing this module raises at import, and every compute
entry point returns an error code when no GPU is bound.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# GRAFT_LIB selects another build of the same library (A/B timing of kernel variants)
LIB_PATH = os.environ.get("GRAFT_LIB") or os.path.join(_HERE, "csrc", "libgrafthip.so")

GRAFT_OK, GRAFT_ERR_ARG, GRAFT_ERR_UNSUPPORTED, GRAFT_ERR_HIP, GRAFT_ERR_STATE = range(5)
GRAFT_SPHERE, GRAFT_PLANE = 0, 1
GRAFT_ORTHOGRAPHIC, GRAFT_PERSPECTIVE = 0, 1
GRAFT_OUT_F64_RGB, GRAFT_OUT_F32_RGBA, GRAFT_OUT_U8_RGBA = 0, 1, 2
DIAGNOSTICS = {"screen_cap": 0, "group_size": 1, "tiles_per_ticket": 2, "table_builds": 4, "reserved_cus": 5, "copy_piece_kb": 6,
               "emulated_devices": 7}

c_double_p = ctypes.POINTER(ctypes.c_double)
c_int32_p = ctypes.POINTER(ctypes.c_int32)


class GraftShape(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int32), ("reserved", ctypes.c_int32), ("geom", ctypes.c_double * 6),
                ("color", ctypes.c_double * 3), ("reflectivity", ctypes.c_double),
                ("ambientLight", ctypes.c_double), ("refrIndex", ctypes.c_double),
                ("transparency", ctypes.c_double)]


class GraftLight(ctypes.Structure):
    _fields_ = [("color", ctypes.c_double * 3), ("loc", ctypes.c_double * 3),
                ("avoidOffset", ctypes.c_int32), ("avoidNum", ctypes.c_int32)]


class GraftScene(ctypes.Structure):
    _fields_ = [("numShapes", ctypes.c_int32), ("numLights", ctypes.c_int32),
                ("shapes", ctypes.POINTER(GraftShape)), ("lights", ctypes.POINTER(GraftLight)),
                ("avoidIndex", c_int32_p), ("avoidIndexLen", ctypes.c_int32), ("maxDepth", ctypes.c_int32),
                ("backgroundColor", ctypes.c_double * 3)]


class GraftView(ctypes.Structure):
    _fields_ = [("screenCenter", ctypes.c_double * 3), ("camPos", ctypes.c_double * 3),
                ("viewDir", ctypes.c_double * 3), ("lrVec", ctypes.c_double * 3), ("udVec", ctypes.c_double * 3),
                ("projectionType", ctypes.c_int32), ("aliasing", ctypes.c_int32), ("numPasses", ctypes.c_int32),
                ("reserved", ctypes.c_int32), ("subStepX", ctypes.c_double), ("subStepY", ctypes.c_double),
                ("nCols", ctypes.c_int32), ("nRows", ctypes.c_int32),
                ("colOffsets", c_double_p), ("rowOffsets", c_double_p)]


class GraftCounters(ctypes.Structure):
    _fields_ = [("rays", ctypes.c_uint64), ("tests", ctypes.c_uint64), ("raysTraced", ctypes.c_uint64),
                ("reserved", ctypes.c_uint64)]


class GraftError(RuntimeError):
    def __init__(self, code, text):
        super().__init__("grafthip error %d: %s" % (code, text))
        self.code = code


if not os.path.exists(LIB_PATH):
    raise ImportError(
        "%s is missing: build it with `make -C raytracing_amd/csrc` (hipcc --offload-arch=gfx950). "
        "There is no CPU fallback for the ray-tracing path." % LIB_PATH)

lib = ctypes.CDLL(LIB_PATH)

lib.graftDeviceCount.restype = ctypes.c_int
lib.graftInit.argtypes = [ctypes.c_int]
lib.graftShutdown.restype = None
lib.graftLastError.restype = ctypes.c_char_p
lib.graftLoopOffsets.argtypes = [ctypes.c_double, ctypes.c_int, c_double_p, ctypes.c_int]
lib.graftScreenCoord.argtypes = [ctypes.c_double, ctypes.c_int, ctypes.c_double]
lib.graftCamInitialize.argtypes = [c_double_p, c_double_p, ctypes.c_double, ctypes.c_double, c_double_p,
                                   c_double_p]
lib.graftCamInitialize.restype = None
lib.graftJobCreate.argtypes = [ctypes.POINTER(GraftScene), ctypes.POINTER(GraftView),
                               ctypes.POINTER(ctypes.c_void_p)]
lib.graftJobDestroy.argtypes = [ctypes.c_void_p]
lib.graftJobDestroy.restype = None
lib.graftJobScreenLevel.argtypes = [ctypes.c_void_p]
lib.graftJobTrace.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                              ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
lib.graftSynchronize.restype = ctypes.c_int
lib.graftRenderRows.argtypes = [ctypes.POINTER(GraftScene), ctypes.POINTER(GraftView), ctypes.c_int,
                                ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
lib.graftRenderRowsNoAlloc.argtypes = [ctypes.POINTER(GraftScene), ctypes.POINTER(GraftView), ctypes.c_int,
                                       ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                       ctypes.c_void_p]
lib.graftInitDevices.argtypes = [ctypes.c_int]
lib.graftFramePieces.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, c_int32_p, c_int32_p, c_int32_p, ctypes.c_int]
if hasattr(lib, "graftFramePlan"):  # (an older build loaded through GRAFT_LIB for A/B timing may lack these two)
    lib.graftFramePlan.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, c_int32_p, c_int32_p, c_int32_p, c_int32_p,
                                   c_int32_p, ctypes.c_int]
    lib.graftMultiStats.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float)]
lib.graftRenderFrameMulti.argtypes = [ctypes.POINTER(GraftScene), ctypes.POINTER(GraftView), ctypes.c_int, ctypes.c_int,
                                      ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
lib.graftWritePPM.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
lib.graftWritePNG.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
lib.graftDiagnosticSet.argtypes = [ctypes.c_int, ctypes.c_int]
lib.graftDiagnosticGet.argtypes = [ctypes.c_int]
lib.graftHostAlloc.argtypes = [ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p)]
lib.graftHostFree.argtypes = [ctypes.c_void_p]
lib.graftHostFree.restype = None
lib.graftPixelBytes.argtypes = [ctypes.c_int]
lib.graftPixelBytes.restype = ctypes.c_size_t
lib.graftProbeSqrt.argtypes = [c_double_p, c_double_p, ctypes.c_size_t]
lib.graftProbeDiv.argtypes = [c_double_p, c_double_p, c_double_p, ctypes.c_size_t]
lib.graftProbeUnit.argtypes = [c_double_p, c_double_p, ctypes.c_size_t]


def check(code):
    if code != GRAFT_OK:
        raise GraftError(code, lib.graftLastError().decode("utf-8", "replace"))
