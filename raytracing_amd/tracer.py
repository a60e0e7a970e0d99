"""Host-side mirror of the reference program around launchRays().

`RayTracer` carries the reference's file-scope globals under their own names (mainTracing.c:29-67) and the
host functions that mutate them (sceneInitialize, camInitialize, rotateView, handleKeyDown, sphereSetup,
planeSetup ...), so a test or script reads like the reference's `main`.  `launchRays()` is the boundary: it
packs the globals into the C ABI structs of include/grafthip.h, runs the HIP path, and feeds the frame to a
pixSetRGB sink in the reference's call order.  All arithmetic done here is the reference's own, in IEEE
double (Python floats), so that what reaches the GPU is bit-for-bit what the C host would produce.
"""
import ctypes
import math

import numpy as np

from . import _lib
from ._lib import (GRAFT_ORTHOGRAPHIC, GRAFT_OUT_F32_RGBA, GRAFT_OUT_F64_RGB, GRAFT_OUT_U8_RGBA,
                   GRAFT_PERSPECTIVE, GRAFT_PLANE, GRAFT_SPHERE, GraftCounters, GraftLight, GraftScene,
                   GraftShape, GraftView, check, lib)

ORTHOGRAPHIC, PERSPECTIVE = GRAFT_ORTHOGRAPHIC, GRAFT_PERSPECTIVE  # mainTracing.c:23-24
ANTIALIASING_OFF, ANTIALIASING_ON = 0, 1                           # mainTracing.c:26-27

_FORMAT_DTYPE = {GRAFT_OUT_F64_RGB: (np.float64, 3), GRAFT_OUT_F32_RGBA: (np.float32, 4),
                 GRAFT_OUT_U8_RGBA: (np.uint8, 4)}

_bound_device = None


def init(device=0):
    """Bind this process to one GPU (graftInit).  Fails loudly when there is none."""
    global _bound_device
    if _bound_device == device:
        return
    check(lib.graftInit(int(device)))
    _bound_device = device


def init_devices(n):
    """Bind GPUs 0..n-1 to this process (graftInitDevices); GPU 0 is the root and the device of the other calls."""
    global _bound_device
    check(lib.graftInitDevices(int(n)))
    _bound_device = 0


def frame_pieces(n_rows, n_devices, pieces_per_device):
    """graftFramePieces: list of (first row, rows, owner device)."""
    cap = lib.graftFramePieces(int(n_rows), int(n_devices), int(pieces_per_device), None, None, None, 0)
    if cap < 0:
        raise ValueError("bad frame or device count")
    a, b, c = ((ctypes.c_int32 * max(cap, 1))() for _ in range(3))
    lib.graftFramePieces(int(n_rows), int(n_devices), int(pieces_per_device), a, b, c, cap)
    return [(a[i], b[i], c[i]) for i in range(cap)]


def frame_plan(n_rows, n_devices, pieces_per_device):
    """graftFramePlan: (pieces, owned rows per device); a piece is (first row, rows, owner device, first row inside
    the owner's own buffer)."""
    cap = lib.graftFramePlan(int(n_rows), int(n_devices), int(pieces_per_device), None, None, None, None, None, 0)
    if cap < 0:
        raise ValueError("bad frame or device count")
    a, b, c, d = ((ctypes.c_int32 * max(cap, 1))() for _ in range(4))
    owned = (ctypes.c_int32 * int(n_devices))()
    lib.graftFramePlan(int(n_rows), int(n_devices), int(pieces_per_device), a, b, c, d, owned, cap)
    return [(a[i], b[i], c[i], d[i]) for i in range(cap)], list(owned)


def multi_stats(device):
    """graftMultiStats: (ms from the device's first kernel to its last, ms from its last kernel to its last transfer)
    of the last multi-device frame."""
    t, tail = ctypes.c_float(), ctypes.c_float()
    check(lib.graftMultiStats(int(device), ctypes.byref(t), ctypes.byref(tail)))
    return t.value, tail.value


def device_count():
    n = lib.graftDeviceCount()
    return max(n, 0)


def set_diagnostic(name, value):
    """graftDiagnosticSet: "screen_cap" 0..7, "group_size" 0|4..64, "tiles_per_ticket", "reserved_cus" (A/B timing,
    tests, and room for transfer kernels beside a trace; no setting changes a result).  Returns the previous value."""
    which = _lib.DIAGNOSTICS[name]
    old = lib.graftDiagnosticGet(which)
    check(lib.graftDiagnosticSet(which, int(value)))
    return old


def get_diagnostic(name):
    return lib.graftDiagnosticGet(_lib.DIAGNOSTICS[name])


def write_ppm(path, frame, fmt=GRAFT_OUT_F64_RGB):
    """Save a host frame ([rows][cols][channels], row 0 at the bottom as pixSetRGB counts) as a binary PPM."""
    a = np.ascontiguousarray(frame)
    check(lib.graftWritePPM(str(path).encode(), a.ctypes.data_as(ctypes.c_void_p), int(fmt), a.shape[1], a.shape[0]))


def write_png(path, frame, fmt=GRAFT_OUT_F64_RGB):
    """Save a host frame ([rows][cols][channels], row 0 at the bottom as pixSetRGB counts) as an 8-bit RGB PNG."""
    a = np.ascontiguousarray(frame)
    check(lib.graftWritePNG(str(path).encode(), a.ctypes.data_as(ctypes.c_void_p), int(fmt), a.shape[1], a.shape[0]))


# ---- shapes.c / light.c ---------------------------------------------------------------------------------

class Shape:
    """`shape` of shapes.c:14-24: unif array + four material scalars; `kind` stands for the two function
    pointers (sphereIntersect/sphereColor or planeIntersect/planeColor)."""

    def __init__(self, kind, unif_len):
        self.kind = kind
        self.unif = [0.0] * unif_len
        self.unifDim = 0
        self.reflectivity = 0.0
        self.ambientLight = 0.0
        self.refrIndex = 0.0
        self.transparency = 0.0


def sphereMalloc():  # shapes.c:125-129
    return Shape(GRAFT_SPHERE, 7)


def planeMalloc():  # shapes.c:104-108
    return Shape(GRAFT_PLANE, 9)


def sphereInit(s, center, radius, reflectivity, colors, ambient, refractionIndex, transparence):  # shapes.c:138-150
    s.kind = GRAFT_SPHERE
    s.ambientLight = float(ambient)
    s.unifDim = 4
    s.unif[0:3] = [float(c) for c in center]
    s.unif[3] = float(radius)
    s.reflectivity = float(reflectivity)
    s.unif[4:7] = [float(c) for c in colors]
    s.refrIndex = float(refractionIndex)
    s.transparency = float(transparence)


def planeInit(p, point, normal, reflectivity, rgb, ambient, refractionIndex, transparence):  # shapes.c:111-122
    p.kind = GRAFT_PLANE
    p.ambientLight = float(ambient)
    p.unif[0:3] = [float(c) for c in point]
    p.unif[3:6] = [float(c) for c in normal]
    p.unif[6:9] = [float(c) for c in rgb]
    p.reflectivity = float(reflectivity)
    p.refrIndex = float(refractionIndex)
    p.transparency = float(transparence)


class Light:
    """`light` of light.c:1-6; `avoid` holds Shape objects, compared by identity as contains() does."""

    def __init__(self):
        self.color = [0.0, 0.0, 0.0]
        self.loc = [0.0, 0.0, 0.0]
        self.avoid = []
        self.avoidNum = 0


def lightInit(l, color, loc):  # light.c:13-18
    l.color = [float(c) for c in color]
    l.loc = [float(c) for c in loc]
    l.avoid = []
    l.avoidNum = 0


def lightSetAvoid(l, avoidList, avoidListNum):  # light.c:8-11
    l.avoid = list(avoidList[:avoidListNum])
    l.avoidNum = int(avoidListNum)


# ---- vector.c / matrix.c pieces the host side uses -------------------------------------------------------

def _vecUnit(v):  # vector.c:74-82 (in place on a copy; unchanged when the length is not positive)
    acc = 0.0
    for c in v:
        acc += c * c
    mag = math.sqrt(acc)
    if mag > 0:
        return [c / mag for c in v]
    return list(v)


def _mat33AngleAxisRotation(theta, axis):  # matrix.c:118-133
    sint, cost = math.sin(theta), math.cos(theta)
    x, y, z = axis
    return [[cost + x * x * (1 - cost), x * y * (1 - cost) - z * sint, x * z * (1 - cost) + y * sint],
            [y * x * (1 - cost) + z * sint, cost + y * y * (1 - cost), y * z * (1 - cost) - x * sint],
            [z * x * (1 - cost) - y * sint, z * y * (1 - cost) + x * sint, cost + z * z * (1 - cost)]]


def _mat331Multiply(m, v):  # matrix.c:72-80
    out = []
    for i in range(3):
        result = 0.0
        for j in range(3):
            result = result + m[i][j] * v[j]
        out.append(result)
    return out


def _d3(v):
    return (ctypes.c_double * 3)(*[float(c) for c in v])


# ---- device-resident job ---------------------------------------------------------------------------------

class Job:
    """A packed scene + view living in GPU memory (graftJobCreate); trace() launches the kernel."""

    def __init__(self, scene_struct, view_struct, keepalive):
        self._h = ctypes.c_void_p()
        check(lib.graftJobCreate(ctypes.byref(scene_struct), ctypes.byref(view_struct), ctypes.byref(self._h)))
        self.nCols, self.nRows = view_struct.nCols, view_struct.nRows
        del keepalive  # the library copied everything it needs

    def trace(self, row0, n_rows, fmt, frame_ptr, hit_index_ptr=None, counters_ptr=None, stream=None):
        check(lib.graftJobTrace(self._h, int(row0), int(n_rows), int(fmt), ctypes.c_void_p(frame_ptr),
                                ctypes.c_void_p(hit_index_ptr or 0), ctypes.c_void_p(counters_ptr or 0),
                                ctypes.c_void_p(stream or 0)))

    def screen_level(self):
        # 0..4: which screen sits in front of the exact sphere test (graftJobScreenLevel)
        return lib.graftJobScreenLevel(self._h)

    def close(self):
        if self._h:
            lib.graftJobDestroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- the program state ---------------------------------------------------------------------------------

class RayTracer:
    def __init__(self, WIDTH=512, HEIGHT=512):
        self.WIDTH, self.HEIGHT = int(WIDTH), int(HEIGHT)          # mainTracing.c:20-21
        self.projectionType = PERSPECTIVE                          # :29, set in main :634
        self.aliasing = ANTIALIASING_ON                            # :30, :635
        self.numPasses = 6                                         # :31, :637
        self.screenHeight = 512.0                                  # :32
        self.screenWidth = 512.0                                   # :33
        self.numShapes = 0                                         # :35
        self.shapes = []                                           # :36
        self.numLights = 0                                         # :38
        self.lights = []                                           # :39
        self.target = [0.0, 0.0, 0.0]                              # :54
        self.targetToScreen = 0.0
        self.screenToCam = 0.0
        self.viewDir = [0.0, 0.0, -1.0]
        self.screenCenter = [0.0, 0.0, 0.0]
        self.camPos = [0.0, 0.0, 0.0]
        self.lrVec = [1.0, 0.0, 0.0]
        self.udVec = [0.0, 1.0, 0.0]
        self.backgroundColor = [0.0, 0.0, 0.0]                     # :63
        self.maxDepth = 4                                          # :64
        self.maxDepthLow = 4
        self.maxDepthHigh = 10
        self.ambientLight = 0.1                                    # :67
        self.frame = np.zeros((self.HEIGHT, self.WIDTH, 3))        # what the display would show (pixSetRGB sink)
        self.last_counters = None      # (closest-hit queries, ray-shape evaluations) of the reference for the last trace
        self.last_rays_traced = None   # closest-hit queries the library made for it

    # -- scene construction (mainTracing.c:447-548) --
    def _place(self, index, s):
        while len(self.shapes) <= index:
            self.shapes.append(None)
        self.shapes[index] = s

    def sphereSetup(self, radius, center, shapeIndex, color, reflectivity, ambient, refrIndex, transparency):  # :460-464
        s = sphereMalloc()
        sphereInit(s, center, radius, reflectivity, color, ambient, refrIndex, transparency)
        self._place(shapeIndex, s)

    def planeSetup(self, normal, center, shapeIndex, color, reflectivity, ambient, refrIndex, transparency):  # :467-471
        p = planeMalloc()
        planeInit(p, center, normal, reflectivity, color, ambient, refrIndex, transparency)
        self._place(shapeIndex, p)

    def sceneInitialize(self, targetPos, targetToScreenDist, screenToCamDist):  # :447-457
        self.viewDir = _vecUnit([0.0, 0.0, -1.0])
        self.lrVec = _vecUnit([1.0, 0.0, 0.0])
        self.udVec = _vecUnit([0.0, 1.0, 0.0])
        self.camInitialize(targetPos, targetToScreenDist, screenToCamDist)

    def camInitialize(self, targetPos, targetToScreenDist, screenToCamDist):  # :375-384
        self.target = [float(c) for c in targetPos]
        self.targetToScreen = float(targetToScreenDist)
        self.screenToCam = float(screenToCamDist)
        sc, cp = (ctypes.c_double * 3)(), (ctypes.c_double * 3)()
        lib.graftCamInitialize(_d3(self.target), _d3(self.viewDir), self.targetToScreen, self.screenToCam, sc, cp)
        self.screenCenter, self.camPos = list(sc), list(cp)

    def initializeShapes(self):
        """The reference's start-up scene: 4 spheres, a closed box of 6 planes, the light's sphere
        (values of mainTracing.c:474-534)."""
        amb = self.ambientLight
        self.shapes = []
        self.sphereSetup(90, [160, -256 + 90, 0], 0, [.2, .8, .1], 0.5, amb, 1.0, 0.0)
        self.sphereSetup(110, [-100, -256 + 110, -100], 1, [.5, .8, .8], 1.0, 0.0, 1.0, 0.0)
        self.sphereSetup(70, [0, -256 + 70, 185], 2, [.9, 0.1, 0.1], .5, amb, 1, 0)
        self.sphereSetup(30, [0, 50, 0], 3, [.8, .8, .8], 1, 0, 1, 0)
        grey = [.5, .5, .5]
        self.planeSetup([0, 1, 0], [0, -256, 0], 4, grey, .2, amb, 1, 0)
        self.planeSetup([1, 0, 0], [-500, 0, 0], 5, grey, .2, amb, 1, 0)
        self.planeSetup([0, 0, 1], [0, 0, -500], 6, grey, .5, amb, 1, 0)
        self.planeSetup([0, 0, -1], [0, 0, 500], 7, grey, .5, amb, 1, 0)
        self.planeSetup([0, -1, 0], [0, 500, 0], 8, grey, .5, amb, 1, 0)
        self.planeSetup([-1, 0, 0], [500, 0, 0], 9, grey, .5, amb, 1, 0)
        light_sphere = sphereMalloc()
        sphereInit(light_sphere, [0, -256 + 256 + 499, 0], 100, 0, [1, 1, 1], 2, 1, 0)
        self._place(10, light_sphere)
        self.numShapes = 11
        self.sceneInitialize([0, 0, 0], 256, 1000)

    def initializeLights(self):  # mainTracing.c:537-548; numLights stays 1, the second light is never used
        l0, l1 = Light(), Light()
        lightInit(l0, [2, 2, 2], [0, -256 + 256 + 499, 0])
        lightSetAvoid(l0, self.shapes[10:11], 1)
        lightInit(l1, [.6, .6, .6], [100, 256, 100])
        self.lights = [l0, l1]
        self.numLights = 1

    # -- interaction (mainTracing.c:71-86, :552-631) --
    def rotateView(self, theta, axis):
        rot = _mat33AngleAxisRotation(theta, axis)
        view, lr, ud = list(self.viewDir), list(self.lrVec), list(self.udVec)
        self.viewDir = _vecUnit(_mat331Multiply(rot, view))
        self.lrVec = _vecUnit(_mat331Multiply(rot, lr))
        self.udVec = _vecUnit(_mat331Multiply(rot, ud))

    def handleKeyDown(self, key, shiftIsDown=0, controlIsDown=0, altOptionIsDown=0, superCommandIsDown=0):
        """mainTracing.c:552-631: change the globals, clear the display, trace a frame."""
        if not self.applyKey(key, shiftIsDown):
            return None
        self.frame[:] = np.asarray(self.backgroundColor)  # pixClearRGB, :629
        return self.launchRays()

    def applyKey(self, key, shiftIsDown=0):
        """The state change of handleKeyDown alone (the switch at mainTracing.c:556-628); False for keys the
        reference ignores."""
        adjustTheta = 3.14 / 16
        udVector = [0.0, 1.0, 0.0]
        if key == 257 or key == 65:  # ENTER falls through into the 'a' case in the reference (:557-569)
            if key == 257:
                self.projectionType = PERSPECTIVE if self.projectionType == ORTHOGRAPHIC else ORTHOGRAPHIC
            self.aliasing = ANTIALIASING_OFF if self.aliasing == ANTIALIASING_ON else ANTIALIASING_ON
        elif key == 262:
            self.rotateView(-adjustTheta, udVector)
        elif key == 263:
            self.rotateView(adjustTheta, udVector)
        elif key == 264:
            if shiftIsDown:
                self.screenWidth *= .9
                self.screenHeight *= .9
            else:
                self.rotateView(-adjustTheta, list(self.lrVec))
        elif key == 265:
            if shiftIsDown:
                self.screenWidth *= 1.1
                self.screenHeight *= 1.1
            else:
                self.rotateView(adjustTheta, list(self.lrVec))
        elif 49 <= key <= 57:
            self.numPasses = key - 48
        elif key == 69:
            self.maxDepth = self.maxDepthHigh if self.maxDepth == self.maxDepthLow else self.maxDepthLow
        else:
            return False
        return True

    def getScreenCoordX(self, coord):  # :89-93
        return lib.graftScreenCoord(float(coord), self.WIDTH, float(self.screenWidth))

    def getScreenCoordY(self, coord):  # :96-100
        return lib.graftScreenCoord(float(coord), self.HEIGHT, float(self.screenHeight))

    # -- the boundary --
    def loop_offsets(self):
        """Values taken by i and j in the loop headers of launchRays (mainTracing.c:392-393)."""
        out = []
        for extent, n in ((self.screenWidth, self.WIDTH), (self.screenHeight, self.HEIGHT)):
            cap = 2 * n + 32
            buf = (ctypes.c_double * cap)()
            count = lib.graftLoopOffsets(float(extent), n, buf, cap)
            if count < 0:
                raise ValueError("loop over extent %r never terminates" % (extent,))
            out.append(np.array(buf[:count], dtype=np.float64))
        return out

    def pack(self):
        """Globals -> (GraftScene, GraftView, keepalive).  Function pointers become kind tags, avoid pointers
        become shape indices by identity."""
        n = int(self.numShapes)
        shapes = (GraftShape * max(n, 1))()
        for k in range(n):
            s = self.shapes[k]
            g = shapes[k]
            g.kind = s.kind
            if s.kind == GRAFT_SPHERE:
                g.geom[0:4] = s.unif[0:4]
                g.color[0:3] = s.unif[4:7]
            else:
                g.geom[0:6] = s.unif[0:6]
                g.color[0:3] = s.unif[6:9]
            g.reflectivity, g.ambientLight = s.reflectivity, s.ambientLight
            g.refrIndex, g.transparency = s.refrIndex, s.transparency
        m = int(self.numLights)
        lights = (GraftLight * max(m, 1))()
        avoid = []
        for i in range(m):
            l = self.lights[i]
            lights[i].color[0:3] = l.color
            lights[i].loc[0:3] = l.loc
            lights[i].avoidOffset = len(avoid)
            for a in l.avoid[:l.avoidNum]:
                idx = -1
                for k in range(n):  # only shapes[0..numShapes) are ever compared, mainTracing.c:133-134
                    if self.shapes[k] is a:
                        idx = k
                        break
                avoid.append(idx)
            lights[i].avoidNum = len(avoid) - lights[i].avoidOffset
        avoid_arr = (ctypes.c_int32 * max(len(avoid), 1))(*avoid)
        scene = GraftScene()
        scene.numShapes, scene.numLights = n, m
        scene.shapes = ctypes.cast(shapes, ctypes.POINTER(GraftShape))
        scene.lights = ctypes.cast(lights, ctypes.POINTER(GraftLight))
        scene.avoidIndex = ctypes.cast(avoid_arr, ctypes.POINTER(ctypes.c_int32))
        scene.avoidIndexLen = len(avoid)
        scene.maxDepth = int(self.maxDepth)
        scene.backgroundColor[0:3] = [float(c) for c in self.backgroundColor]

        cols, rows = self.loop_offsets()
        view = GraftView()
        view.screenCenter[0:3] = self.screenCenter
        view.camPos[0:3] = self.camPos
        view.viewDir[0:3] = self.viewDir
        view.lrVec[0:3] = self.lrVec
        view.udVec[0:3] = self.udVec
        view.projectionType = int(self.projectionType)
        view.aliasing = 1 if self.aliasing == ANTIALIASING_ON else 0
        view.numPasses = int(self.numPasses)
        if view.aliasing:
            view.subStepX = self.screenWidth / self.WIDTH / self.numPasses   # :405
            view.subStepY = self.screenHeight / self.WIDTH / self.numPasses  # :406 (WIDTH, as the reference)
        view.nCols, view.nRows = len(cols), len(rows)
        view.colOffsets = cols.ctypes.data_as(_lib.c_double_p)
        view.rowOffsets = rows.ctypes.data_as(_lib.c_double_p)
        return scene, view, (shapes, lights, avoid_arr, cols, rows)

    def create_job(self):
        """camInitialize (:388) + pack + upload; returns a device-resident Job."""
        init(_bound_device if _bound_device is not None else 0)
        self.camInitialize(self.target, self.targetToScreen, self.screenToCam)
        scene, view, keep = self.pack()
        return Job(scene, view, keep)

    def render_rows(self, row0=0, n_rows=None, fmt=GRAFT_OUT_F64_RGB, want_index=False, want_counters=False):
        """Trace loop rows [row0,row0+n_rows) on the GPU and return host arrays indexed [iy-row0][ix]."""
        init(_bound_device if _bound_device is not None else 0)
        self.camInitialize(self.target, self.targetToScreen, self.screenToCam)
        scene, view, keep = self.pack()
        if n_rows is None:
            n_rows = view.nRows - row0
        dtype, ch = _FORMAT_DTYPE[fmt]
        frame = np.zeros((n_rows, view.nCols, ch), dtype=dtype)
        index = np.full((n_rows, view.nCols), -9, dtype=np.int32) if want_index else None
        counters = GraftCounters() if want_counters else None
        check(lib.graftRenderRows(ctypes.byref(scene), ctypes.byref(view), int(row0), int(n_rows), int(fmt),
                                  frame.ctypes.data_as(ctypes.c_void_p),
                                  index.ctypes.data_as(ctypes.c_void_p) if want_index else None,
                                  ctypes.byref(counters) if want_counters else None))
        del keep
        if want_counters:
            self.last_counters = (counters.rays, counters.tests)
            self.last_rays_traced = counters.raysTraced
        return (frame, index) if want_index else frame

    def render_frame_multi(self, fmt=GRAFT_OUT_F64_RGB, pieces_per_device=8, want_index=False, want_counters=False):
        """graftRenderFrameMulti: the whole frame over every GPU bound by init_devices(), assembled on GPU 0."""
        self.camInitialize(self.target, self.targetToScreen, self.screenToCam)
        scene, view, keep = self.pack()
        dtype, ch = _FORMAT_DTYPE[fmt]
        frame = np.zeros((view.nRows, view.nCols, ch), dtype=dtype)
        index = np.full((view.nRows, view.nCols), -9, dtype=np.int32) if want_index else None
        counters = GraftCounters() if want_counters else None
        check(lib.graftRenderFrameMulti(ctypes.byref(scene), ctypes.byref(view), int(fmt), int(pieces_per_device),
                                        frame.ctypes.data_as(ctypes.c_void_p),
                                        index.ctypes.data_as(ctypes.c_void_p) if want_index else None,
                                        ctypes.byref(counters) if want_counters else None))
        del keep
        if want_counters:
            self.last_counters = (counters.rays, counters.tests)
            self.last_rays_traced = counters.raysTraced
        return (frame, index) if want_index else frame

    def render_rows_no_alloc(self, row0, n_rows, fmt, frame_ptr, index_ptr=None, counters_ptr=None, stream=None):
        """graftRenderRowsNoAlloc: trace loop rows [row0,row0+n_rows) into DEVICE buffers of the caller without any
        device allocation (asynchronous on `stream`, None = the library's stream: lib.graftSynchronize() waits)."""
        init(_bound_device if _bound_device is not None else 0)
        self.camInitialize(self.target, self.targetToScreen, self.screenToCam)
        scene, view, keep = self.pack()
        check(lib.graftRenderRowsNoAlloc(ctypes.byref(scene), ctypes.byref(view), int(row0), int(n_rows), int(fmt),
                                         ctypes.c_void_p(frame_ptr), ctypes.c_void_p(index_ptr or 0),
                                         ctypes.c_void_p(counters_ptr or 0), ctypes.c_void_p(stream or 0)))
        del keep

    def launchRays(self, pixSetRGB=None):
        """mainTracing.c:387-443.  Traces the whole frame on the GPU, then replays the reference's pixSetRGB
        calls (x outer, y inner, :392-393, :439) into self.frame[y][x] -- and into `pixSetRGB(x, y, r, g, b)`
        one call per pixel if a sink is given.  Returns self.frame."""
        rgb = self.render_rows(0, None, GRAFT_OUT_F64_RGB)
        cols, rows = self.loop_offsets()
        px = np.array([self.getScreenCoordX(i) for i in cols], dtype=np.int64)
        py = np.array([self.getScreenCoordY(j) for j in rows], dtype=np.int64)
        if pixSetRGB is not None:
            for a in range(len(cols)):
                for b in range(len(rows)):
                    pixSetRGB(int(px[a]), int(py[b]), rgb[b, a, 0], rgb[b, a, 1], rgb[b, a, 2])
        okx = (px >= 0) & (px < self.WIDTH)
        oky = (py >= 0) & (py < self.HEIGHT)
        ia, ib = np.nonzero(okx)[0], np.nonzero(oky)[0]
        # x-outer / y-inner order so that, where two loop cells land on one pixel, the later call wins
        xs = np.repeat(px[ia], len(ib))
        ys = np.tile(py[ib], len(ia))
        vals = rgb[np.tile(ib, len(ia)), np.repeat(ia, len(ib))]
        flat = ys * self.WIDTH + xs
        _, first_from_end = np.unique(flat[::-1], return_index=True)
        last = len(flat) - 1 - first_from_end
        self.frame.reshape(-1, 3)[flat[last]] = vals[last]
        return self.frame
