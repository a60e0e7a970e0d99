"""ctypes binding of the reference builds under oracle/_ref  (TEST INFRASTRUCTURE ONLY).

The libraries are the real walker-a/RayTracing sources compiled by oracle/Makefile; the functions bound
here are the small setters of oracle/ref_harness_body.c around the reference's own globals.  Each library
holds one set of globals, so one `Reference` object per library and no threads.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_REF_DIR = os.path.join(_HERE, "_ref")

ORTHOGRAPHIC, PERSPECTIVE = 0, 1          # mainTracing.c:23-24
KEY_ENTER, KEY_A, KEY_E = 257, 65, 69     # mainTracing.c:557, :563, :619
KEY_RIGHT, KEY_LEFT, KEY_DOWN, KEY_UP = 262, 263, 264, 265

_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int32)


def _d3(v):
    return (ctypes.c_double * 3)(*[float(x) for x in v])


def available(name="libref_default.so"):
    return os.path.exists(os.path.join(_REF_DIR, name))


def fnv1a64(buf):
    """FNV-1a 64 over the raw bytes of a numpy array (vectorised per 8-byte... no: byte-wise, as specified)."""
    data = np.ascontiguousarray(buf).view(np.uint8).ravel()
    h = 0xCBF29CE484222325
    prime = 0x100000001B3
    mask = (1 << 64) - 1
    for b in data.tobytes():
        h = ((h ^ b) * prime) & mask
    return "%016x" % h


class Reference:
    def __init__(self, size=None):
        """size=None: the unmodified 512x512 build; size=(W, H): the constants-patched build of that size."""
        name = "libref_default.so" if size is None else "librefp_%dx%d.so" % tuple(size)
        path = os.path.join(_REF_DIR, name)
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        L = self.lib = ctypes.CDLL(path)
        L.ref_sink_calls.restype = ctypes.c_long
        L.ref_sink_out_of_range.restype = ctypes.c_long
        L.ref_set_screen.argtypes = [ctypes.c_double, ctypes.c_double]
        L.ref_add_sphere.argtypes = [_dp, ctypes.c_double, ctypes.c_double, _dp, ctypes.c_double,
                                     ctypes.c_double, ctypes.c_double]
        L.ref_add_plane.argtypes = [_dp, _dp, ctypes.c_double, _dp, ctypes.c_double, ctypes.c_double,
                                    ctypes.c_double]
        L.ref_add_light.argtypes = [_dp, _dp, _ip, ctypes.c_int]
        L.ref_set_material.argtypes = [ctypes.c_int] + [ctypes.c_double] * 4
        L.ref_scene_initialize.argtypes = [_dp, ctypes.c_double, ctypes.c_double]
        L.ref_rotate_view.argtypes = [ctypes.c_double, _dp]
        L.ref_set_background.argtypes = [_dp]
        L.ref_get_state.argtypes = [_dp]
        L.ref_launch.argtypes = [_dp, _ip, ctypes.c_long]
        L.ref_key.argtypes = [ctypes.c_int, ctypes.c_int, _dp, _ip, ctypes.c_long]
        L.ref_primary_index.argtypes = [_ip]
        self.W, self.H = L.ref_width(), L.ref_height()
        self.max_shapes, self.max_lights = L.ref_max_shapes(), L.ref_max_lights()

    # -- scene -----------------------------------------------------------------------------------
    def default_scene(self):
        self.lib.ref_default_scene()

    def clear_scene(self):
        self.lib.ref_clear_scene()

    def add_sphere(self, center, radius, reflectivity, rgb, ambient, refr_index=1.0, transparency=0.0):
        r = self.lib.ref_add_sphere(_d3(center), radius, reflectivity, _d3(rgb), ambient, refr_index,
                                    transparency)
        assert r >= 0, "reference shape array full"
        return r

    def add_plane(self, point, normal, reflectivity, rgb, ambient, refr_index=1.0, transparency=0.0):
        r = self.lib.ref_add_plane(_d3(point), _d3(normal), reflectivity, _d3(rgb), ambient, refr_index,
                                   transparency)
        assert r >= 0, "reference shape array full"
        return r

    def add_light(self, color, loc, avoid=()):
        arr = (ctypes.c_int32 * max(1, len(avoid)))(*avoid)
        r = self.lib.ref_add_light(_d3(color), _d3(loc), arr, len(avoid))
        assert r >= 0, "reference light array full"
        return r

    def set_material(self, k, reflectivity, ambient, refr_index, transparency):
        self.lib.ref_set_material(k, reflectivity, ambient, refr_index, transparency)

    def scene_initialize(self, target=(0, 0, 0), to_screen=256.0, to_cam=1000.0):
        self.lib.ref_scene_initialize(_d3(target), to_screen, to_cam)

    def set_flags(self, projection=PERSPECTIVE, aliasing=0, passes=1, depth=4):
        self.lib.ref_set_flags(projection, aliasing, passes, depth)

    def set_counts(self, n_shapes, n_lights):
        self.lib.ref_set_counts(n_shapes, n_lights)

    def set_screen(self, w, h):
        self.lib.ref_set_screen(w, h)

    def set_background(self, rgb):
        self.lib.ref_set_background(_d3(rgb))

    def rotate_view(self, theta, axis):
        self.lib.ref_rotate_view(theta, _d3(axis))

    def state(self):
        out = np.zeros(27)
        self.lib.ref_get_state(out.ctypes.data_as(_dp))
        names = ["target", "viewDir", "lrVec", "udVec", "screenCenter", "camPos"]
        st = {n: out[3 * i:3 * i + 3].copy() for i, n in enumerate(names)}
        st.update(targetToScreen=out[18], screenToCam=out[19], screenWidth=out[20], screenHeight=out[21],
                  projectionType=int(out[22]), aliasing=int(out[23]), numPasses=int(out[24]),
                  maxDepth=int(out[25]), numShapes=int(out[26]))
        return st

    # -- rendering -------------------------------------------------------------------------------
    def _log(self, want_log):
        if not want_log:
            return None, None, 0
        cap = (self.W + 2) * (self.H + 2)
        log = np.full((cap, 2), -12345, dtype=np.int32)
        return log, log.ctypes.data_as(_ip), cap

    def launch(self, fill=0.0, want_log=False):
        """launchRays() into a frame pre-filled with `fill`; returns frame[H][W][3] (and the call log)."""
        frame = np.full((self.H, self.W, 3), fill, dtype=np.float64)
        log, lp, cap = self._log(want_log)
        self.lib.ref_launch(frame.ctypes.data_as(_dp), lp, cap)
        if want_log:
            return frame, log[:self.lib.ref_sink_calls()]
        return frame

    def key(self, key, shift=0, want_log=False):
        """handleKeyDown(key, shift, ...): mutate globals, clear to background, launchRays()."""
        frame = np.zeros((self.H, self.W, 3), dtype=np.float64)
        log, lp, cap = self._log(want_log)
        self.lib.ref_key(key, shift, frame.ctypes.data_as(_dp), lp, cap)
        if want_log:
            return frame, log[:self.lib.ref_sink_calls()]
        return frame

    def primary_index(self, fill=-7):
        idx = np.full((self.H, self.W), fill, dtype=np.int32)
        self.lib.ref_primary_index(idx.ctypes.data_as(_ip))
        return idx

    def out_of_range(self):
        return self.lib.ref_sink_out_of_range()
