"""ctypes binding of oracle/librt_oracle.so, the CPU restatement of the path  (TEST INFRASTRUCTURE ONLY).

`render(state, ...)` takes any object that carries the reference's globals under their own names
(projectionType, aliasing, numPasses, screenWidth, shapes, lights, viewDir, ... -- raytracing_amd.RayTracer
does) and derives everything else (loop offsets, screen centre, camera position) with the oracle's OWN
restatement of the reference's host arithmetic, not with the product's.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librt_oracle.so")

_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int32)


class OracleJob(ctypes.Structure):
    _fields_ = [("numShapes", ctypes.c_int32), ("numLights", ctypes.c_int32),
                ("kind", _ip), ("unif", _dp), ("material", _dp),
                ("lightColor", _dp), ("lightLoc", _dp), ("avoidOff", _ip), ("avoidIdx", _ip),
                ("background", ctypes.c_double * 3),
                ("screenCenter", ctypes.c_double * 3), ("camPos", ctypes.c_double * 3),
                ("viewDir", ctypes.c_double * 3), ("lrVec", ctypes.c_double * 3), ("udVec", ctypes.c_double * 3),
                ("maxDepth", ctypes.c_int32), ("projectionType", ctypes.c_int32),
                ("aliasing", ctypes.c_int32), ("numPasses", ctypes.c_int32),
                ("subStepX", ctypes.c_double), ("subStepY", ctypes.c_double),
                ("nCols", ctypes.c_int32), ("nRows", ctypes.c_int32),
                ("colOff", _dp), ("rowOff", _dp), ("skipLightsBehind", ctypes.c_int32)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError("%s missing: run `make -C oracle librt_oracle.so`" % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        L.oracle_render.argtypes = [ctypes.POINTER(OracleJob), ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                    ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        L.oracle_loop_offsets.argtypes = [ctypes.c_double, ctypes.c_int, _dp, ctypes.c_int]
        L.oracle_screen_coord.argtypes = [ctypes.c_double, ctypes.c_int, ctypes.c_double]
        L.oracle_cam_initialize.argtypes = [_dp, _dp, ctypes.c_double, ctypes.c_double, _dp, _dp]
        L.oracle_cam_initialize.restype = None
        L.oracle_fnv1a64.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
        L.oracle_fnv1a64.restype = ctypes.c_uint64
        _lib = L
    return _lib


def fnv1a64(arr):
    a = np.ascontiguousarray(arr)
    return "%016x" % lib().oracle_fnv1a64(a.ctypes.data_as(ctypes.c_void_p), a.nbytes)


def loop_offsets(extent, n):
    cap = 4 * n + 32
    buf = np.zeros(cap)
    count = lib().oracle_loop_offsets(float(extent), int(n), buf.ctypes.data_as(_dp), cap)
    return buf[:count].copy()


def screen_coord(coord, n, extent):
    return lib().oracle_screen_coord(float(coord), int(n), float(extent))


def cam_initialize(target, view_dir, to_screen, to_cam):
    t, v = np.array(target, dtype=np.float64), np.array(view_dir, dtype=np.float64)
    sc, cp = np.zeros(3), np.zeros(3)
    lib().oracle_cam_initialize(t.ctypes.data_as(_dp), v.ctypes.data_as(_dp), float(to_screen), float(to_cam),
                                sc.ctypes.data_as(_dp), cp.ctypes.data_as(_dp))
    return sc, cp


class Prepared:
    """OracleJob + the numpy arrays it points into."""

    @staticmethod
    def shape_colour(s):
        """R, G, B of a shape: unif[4:7] of a sphere (shapes.c:144-147), unif[6:9] of a plane (:114-116)."""
        return list(s.unif[4:7]) if s.kind == 0 else list(s.unif[6:9])

    def __init__(self, st):
        n, m = int(st.numShapes), int(st.numLights)
        self.kind = np.zeros(max(n, 1), dtype=np.int32)
        self.unif = np.zeros((max(n, 1), 9))
        self.material = np.zeros((max(n, 1), 4))
        for k in range(n):
            s = st.shapes[k]
            self.kind[k] = s.kind
            self.unif[k, :len(s.unif)] = s.unif
            self.material[k] = [s.reflectivity, s.ambientLight, s.refrIndex, s.transparency]
        self.lcol = np.zeros((max(m, 1), 3))
        self.lloc = np.zeros((max(m, 1), 3))
        off, idx = [0], []
        for i in range(m):
            l = st.lights[i]
            self.lcol[i], self.lloc[i] = l.color, l.loc
            for a in l.avoid[:l.avoidNum]:
                hits = [k for k in range(n) if st.shapes[k] is a]
                idx.append(hits[0] if hits else -1)
            off.append(len(idx))
        self.aoff = np.array(off, dtype=np.int32)
        self.aidx = np.array(idx + [0], dtype=np.int32)
        self.cols = loop_offsets(st.screenWidth, st.WIDTH)
        self.rows = loop_offsets(st.screenHeight, st.HEIGHT)
        sc, cp = cam_initialize(st.target, st.viewDir, st.targetToScreen, st.screenToCam)
        J = self.job = OracleJob()
        J.numShapes, J.numLights = n, m
        J.kind = self.kind.ctypes.data_as(_ip)
        J.unif = self.unif.ctypes.data_as(_dp)
        J.material = self.material.ctypes.data_as(_dp)
        J.lightColor = self.lcol.ctypes.data_as(_dp)
        J.lightLoc = self.lloc.ctypes.data_as(_dp)
        J.avoidOff = self.aoff.ctypes.data_as(_ip)
        J.avoidIdx = self.aidx.ctypes.data_as(_ip)
        J.background[0:3] = [float(c) for c in st.backgroundColor]
        J.screenCenter[0:3] = list(sc)
        J.camPos[0:3] = list(cp)
        J.viewDir[0:3] = [float(c) for c in st.viewDir]
        J.lrVec[0:3] = [float(c) for c in st.lrVec]
        J.udVec[0:3] = [float(c) for c in st.udVec]
        J.maxDepth, J.projectionType = int(st.maxDepth), int(st.projectionType)
        J.aliasing, J.numPasses = (1 if st.aliasing else 0), int(st.numPasses)
        if st.aliasing:
            J.subStepX = st.screenWidth / st.WIDTH / st.numPasses
            J.subStepY = st.screenHeight / st.WIDTH / st.numPasses
        J.nCols, J.nRows = len(self.cols), len(self.rows)
        J.colOff = self.cols.ctypes.data_as(_dp)
        J.rowOff = self.rows.ctypes.data_as(_dp)
        colours = np.concatenate([self.lcol[:m].ravel(), np.array([c for k in range(n) for c in self.shape_colour(st.shapes[k])])])
        weights = self.material[:n][:, [0, 1, 3]].ravel()
        J.skipLightsBehind = int(bool(n >= 32 and m <= 31 and np.all(np.isfinite(colours)) and np.all(np.isfinite(weights))
                                      and (colours.size == 0 or np.abs(colours).max() < 1e150)))
        self.WIDTH, self.HEIGHT = int(st.WIDTH), int(st.HEIGHT)
        self.screenWidth, self.screenHeight = float(st.screenWidth), float(st.screenHeight)


def render(st, row0=0, n_rows=None, threads=1, want_rgb=True, want_index=False, want_counters=False):
    """Loop rows [row0,row0+n_rows) -> dict(rgb=[rows][nCols][3], index=[rows][nCols], counters=(rays,tests,shoots,overwritten shadow rays))."""
    pr = Prepared(st)
    J = pr.job
    if n_rows is None:
        n_rows = J.nRows - row0
    rgb = np.zeros((n_rows, J.nCols, 3)) if want_rgb else None
    index = np.full((n_rows, J.nCols), -9, dtype=np.int32) if want_index else None
    cnt = np.zeros(4, dtype=np.uint64) if want_counters else None
    rc = lib().oracle_render(ctypes.byref(J), int(row0), int(row0 + n_rows), int(threads),
                             rgb.ctypes.data_as(ctypes.c_void_p) if want_rgb else None,
                             index.ctypes.data_as(ctypes.c_void_p) if want_index else None,
                             cnt.ctypes.data_as(ctypes.c_void_p) if want_counters else None)
    if rc != 0:
        raise ValueError("oracle_render failed (%d)" % rc)
    out = dict(rgb=rgb, index=index, cols=pr.cols, rows=pr.rows)
    if want_counters:
        out["counters"] = tuple(int(c) for c in cnt)
    return out


def display_frame(st, rgb, cols, rows, fill=0.0):
    """Replay pixSetRGB(getScreenCoordX(i), getScreenCoordY(j), ...) in the reference's call order (x outer,
    y inner) into frame[y][x]; returns (frame, calls_out_of_range)."""
    W, H = int(st.WIDTH), int(st.HEIGHT)
    frame = np.full((H, W, 3), fill, dtype=np.float64)
    px = [screen_coord(i, W, st.screenWidth) for i in cols]
    py = [screen_coord(j, H, st.screenHeight) for j in rows]
    oob = 0
    for a, x in enumerate(px):
        for b, y in enumerate(py):
            if 0 <= x < W and 0 <= y < H:
                frame[y, x] = rgb[b, a]
            else:
                oob += 1
    return frame, oob
