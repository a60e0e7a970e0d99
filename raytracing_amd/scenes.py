"""Scenes the path is measured and tested on.

`default_scene` is the reference program's own start-up state (mainTracing.c:633-641).  `synthetic_scene`
builds the random-sphere workloads: N spheres in a box in front of the screen plane and four white lights,
from a SplitMix64 stream, so that every backend (reference build, CPU checker, GPU) sees identical doubles.
"""
from .tracer import (ANTIALIASING_OFF, ANTIALIASING_ON, PERSPECTIVE, Light, RayTracer, lightInit)

_MASK = (1 << 64) - 1


class SplitMix64:
    def __init__(self, state):
        self.state = state & _MASK

    def next(self):
        self.state = (self.state + 0x9E3779B97F4A7C15) & _MASK
        z = self.state
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _MASK
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _MASK
        return z ^ (z >> 31)

    def uniform(self, a, b):
        return a + (b - a) * ((self.next() >> 11) * 2.0 ** -53)


# name -> (frame W, H, spheres, radius range, reflectivity, maxDepth, aliasing, numPasses, generator seed tag)
WORKLOADS = {
    "cfg3": dict(W=4096, H=4096, n=256, rlo=8.0, rhi=30.0, refl=0.0, depth=4, passes=0, tag=3),
    "cfg4": dict(W=16384, H=16384, n=1024, rlo=5.0, rhi=19.0, refl=0.0, depth=4, passes=4, tag=4),
    "cfg5": dict(W=8192, H=8192, n=4096, rlo=3.0, rhi=12.0, refl=0.5, depth=2, passes=0, tag=5),
}

LIGHT_POSITIONS = [(300.0, 450.0, 300.0), (-300.0, 450.0, 300.0), (300.0, 450.0, -300.0), (-300.0, 450.0, -300.0)]


def default_scene(aliasing=ANTIALIASING_ON, numPasses=6, maxDepth=4, projectionType=PERSPECTIVE, WIDTH=512,
                  HEIGHT=512):
    rt = RayTracer(WIDTH, HEIGHT)
    rt.projectionType, rt.aliasing, rt.numPasses, rt.maxDepth = projectionType, aliasing, numPasses, maxDepth
    rt.initializeShapes()
    rt.initializeLights()
    return rt


def single_sphere_scene(**kw):
    """The start-up scene cut down to shape 0 and light 0 (numShapes = 1)."""
    rt = default_scene(**kw)
    rt.numShapes = 1
    return rt


def synthetic_scene(n, rlo, rhi, refl, depth, tag, W, H, passes=0, lights=4):
    rt = RayTracer(W, H)
    rng = SplitMix64(0x5EED0000 + tag)
    for k in range(n):
        cx = rng.uniform(-450, 450)
        cy = rng.uniform(-450, 450)
        cz = rng.uniform(-200, 450)
        r = rng.uniform(rlo, rhi)
        rgb = [rng.uniform(0.1, 1), rng.uniform(0.1, 1), rng.uniform(0.1, 1)]
        rt.sphereSetup(r, [cx, cy, cz], k, rgb, refl, 0.1, 1.0, 0.0)
    rt.numShapes = n
    rt.lights = []
    for pos in LIGHT_POSITIONS[:lights]:
        l = Light()
        lightInit(l, [1, 1, 1], pos)
        rt.lights.append(l)
    rt.numLights = len(rt.lights)
    rt.maxDepth = depth
    rt.aliasing = ANTIALIASING_ON if passes else ANTIALIASING_OFF
    rt.numPasses = passes if passes else 1
    rt.projectionType = PERSPECTIVE
    rt.sceneInitialize([0, 0, 0], 256, 1000)
    return rt


def workload(name, W=None, H=None):
    """One of the named synthetic workloads, optionally at a reduced frame size (same scene, same window)."""
    w = WORKLOADS[name]
    return synthetic_scene(w["n"], w["rlo"], w["rhi"], w["refl"], w["depth"], w["tag"], W or w["W"], H or w["H"],
                           w["passes"])
