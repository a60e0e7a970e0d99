"""Grazing rays: spheres placed tangent to primary and shadow rays to within a few units in the last place of the
discriminant b*b - (v.v - r*r) (shapes.c:54-57), where the conservative screens in front of the exact test have the
least room; at ordinary coordinates, at coordinates just under the limits of the two-level screens and of the grid
walk, and with lights and the camera a hair outside a sphere (where a cone table entry turns into "always passes").
GPU against the CPU oracle, bit for bit, at every screen level.

The rays are reproduced here in Python floats with the reference's own operations in the reference's order
(getNewPoint mainTracing.c:104-114, vecUnit vector.c:74-82, sphereIntersect shapes.c:47-73, lighting
mainTracing.c:199-205), so that the tangent spheres sit on the rays the tracer really shoots."""
import math

import numpy as np
import pytest

import oracle_binding as ob
from raytracing_amd import get_diagnostic, scenes, set_diagnostic
from raytracing_amd.tracer import ANTIALIASING_OFF, GRAFT_OUT_F64_RGB, ORTHOGRAPHIC, Light, RayTracer, lightInit

pytestmark = pytest.mark.gpu


def dot(v, w):
    acc = 0.0
    for a, b in zip(v, w):
        acc += a * b
    return acc


def unit(v):
    mag = math.sqrt(dot(v, v))
    return [x / mag for x in v] if mag > 0 else list(v)


def new_point(start, direction, dist):
    den = direction[0] * direction[0] + direction[1] * direction[1] + direction[2] * direction[2]
    sc = math.sqrt(dist * dist / den)
    if dist < 0:
        sc = -sc
    return [s + sc * d for s, d in zip(start, direction)]


def primary_ray(rt, i, j):
    fin = new_point(new_point(rt.screenCenter, rt.lrVec, i), rt.udVec, j)
    if rt.projectionType == ORTHOGRAPHIC:   # mainTracing.c:410-413
        return fin, unit([-a for a in rt.viewDir])
    return fin, unit([a - b for a, b in zip(fin, rt.camPos)])


def disc(s, d, c, r):
    v = [a - b for a, b in zip(s, c)]
    b = dot(v, d)
    return b * b - (dot(v, v) - r * r)


def sphere_hit(s, d, c, r):
    v = [a - b for a, b in zip(s, c)]
    b = dot(v, d)
    q = b * b - (dot(v, v) - r * r)
    if q < 0:
        return None
    q = math.sqrt(q)
    t = (-b - q) if (-b - q) >= 0 else (-b + q)
    if t < 0:
        return None
    return [a + t * e for a, e in zip(s, d)]


def perpendicular(d):
    axis = [1.0, 0.0, 0.0] if abs(d[0]) < 0.6 else [0.0, 1.0, 0.0]
    c = [d[1] * axis[2] - d[2] * axis[1], d[2] * axis[0] - d[0] * axis[2], d[0] * axis[1] - d[1] * axis[0]]
    return unit(c)


def tangent_radii(s, d, centre, r):
    """Radii around r at which the discriminant of (s, d) changes sign: the last radius with a negative value moved by
    -4, -1, 0, +1, +4 units in the last place."""
    lo, hi = r * (1 - 1e-9), r * (1 + 1e-9)
    assert disc(s, d, centre, lo) < 0 <= disc(s, d, centre, hi)
    for _ in range(80):
        mid = 0.5 * (lo + hi)
        if mid == lo or mid == hi:
            break
        if disc(s, d, centre, mid) < 0:
            lo = mid
        else:
            hi = mid
    out = []
    for k in (-4, -1, 0, 1, 4):
        x = lo
        for _ in range(abs(k)):
            x = math.nextafter(x, math.inf if k > 0 else -math.inf)
        out.append(x)
    return out


def grazing_scene(offset, fillers, seed, size=(64, 64), mirrors=True, projection=None):
    """A floor sphere, two lights, tangent spheres on the primary rays of a lattice of pixels and on the shadow rays
    of another lattice, plus `fillers` small random spheres; everything moved by `offset`.  Without `mirrors` no shape
    reflects (the candidate lists of level 6 take such scenes)."""
    ox, oy, oz = offset
    rt = RayTracer(*size)
    rt.aliasing, rt.numPasses, rt.maxDepth = ANTIALIASING_OFF, 1, 3
    if projection is not None:
        rt.projectionType = projection
    rt.sceneInitialize([ox, oy, oz], 256, 1000)
    rt.camInitialize(rt.target, rt.targetToScreen, rt.screenToCam)
    for pos in ((ox + 40.0, oy + 420.0, oz + 60.0), (ox - 260.0, oy + 380.0, oz - 120.0)):
        l = Light()
        lightInit(l, [1.0, 0.9, 0.8], list(pos))
        rt.lights.append(l)
    rt.numLights = 2
    n = 0

    def add(c, r, refl=0.0):
        nonlocal n
        rt.sphereSetup(r, c, n, [.3 + .07 * (n % 10), .9 - .06 * (n % 11), .5], refl if mirrors else 0.0, 0.1, 1.0, 0.0)
        n += 1

    floor_c, floor_r = [ox, oy - 1180.0, oz + 150.0], 1000.0
    add(floor_c, floor_r, 0.3)
    cols, rows = rt.loop_offsets()
    # tangent to primary rays
    for a in range(4, size[0], 9):
        for b in range(size[1] // 2 + 3, size[1], 9):   # upper half: these rays miss the floor
            s, d = primary_ray(rt, float(cols[a]), float(rows[b]))
            t0, r = 900.0 + 13.0 * ((a + b) % 7), 4.0 + 0.37 * ((a * 7 + b) % 9)
            nrm = perpendicular(d)
            centre = [s[c] + t0 * d[c] + r * nrm[c] for c in range(3)]
            for rr in tangent_radii(s, d, centre, r)[(a + b) % 5::5] or [r]:
                add(centre, rr)
    # tangent to shadow rays from floor hits towards light 0
    L = rt.lights[0].loc
    for a in range(6, size[0], 11):
        for b in range(2, size[1] // 2 - 6, 7):
            s, d = primary_ray(rt, float(cols[a]), float(rows[b]))
            loc = sphere_hit(s, d, floor_c, floor_r)
            if loc is None:
                continue
            sd = unit([L[c] - loc[c] for c in range(3)])
            so = [loc[c] + .000001 * sd[c] for c in range(3)]
            t0, r = 120.0 + 9.0 * ((a + b) % 5), 3.0 + 0.41 * ((a + 3 * b) % 7)
            nrm = perpendicular(sd)
            centre = [so[c] + t0 * sd[c] + r * nrm[c] for c in range(3)]
            for rr in tangent_radii(so, sd, centre, r)[(a * 3 + b) % 5::5] or [r]:
                add(centre, rr)
    rng = scenes.SplitMix64(0x6A2E0000 + seed)
    for _ in range(fillers):
        add([ox + rng.uniform(-420, 420), oy + rng.uniform(-150, 400), oz + rng.uniform(-100, 420)], rng.uniform(1.5, 6.0),
            0.4 if rng.uniform(0, 1) < 0.3 else 0.0)
    rt.numShapes = n
    return rt


def check_levels(rt, levels):
    want = ob.render(rt, threads=16, want_index=True)
    old = get_diagnostic("screen_cap")
    used = []
    try:
        for level in levels:
            set_diagnostic("screen_cap", level)
            job = rt.create_job()
            used.append(job.screen_level())
            job.close()
            rgb, idx = rt.render_rows(0, None, GRAFT_OUT_F64_RGB, want_index=True)
            assert np.array_equal(idx, want["index"]), "index mask differs at screen cap %d" % level
            assert np.array_equal(rgb.view(np.uint64), want["rgb"].view(np.uint64)), "RGB differs at screen cap %d" % level
    finally:
        set_diagnostic("screen_cap", old)
    return used, want


def test_tangent_spheres_ordinary_coordinates(gpu):
    used, want = check_levels(grazing_scene((0.0, 0.0, 0.0), 60, 1), (0, 1, 2, 3))
    assert used == [0, 1, 2, 3]
    assert (want["index"] > 0).sum() > 20   # tangent spheres and fillers are really seen
    used, _ = check_levels(grazing_scene((0.0, 0.0, 0.0), 2300, 2, size=(48, 48)), (0, 4, 5, 7))
    assert used == [0, 4, 5, 7]
    # the candidate lists: tile lists against tangent primary rays, occluder lists against tangent shadow rays
    used, want = check_levels(grazing_scene((0.0, 0.0, 0.0), 60, 1, mirrors=False), (0, 6))
    assert used == [0, 6] and (want["index"] > 0).sum() > 20
    used, _ = check_levels(grazing_scene((0.0, 0.0, 0.0), 60, 1), (0, 7))
    assert used == [0, 7]
    used, _ = check_levels(grazing_scene((0.0, 0.0, 0.0), 2300, 2, size=(48, 48), mirrors=False), (0, 7))
    assert used == [0, 7]


def test_tangent_spheres_orthographic(gpu):
    """The same lattice of tangent spheres on the parallel primary rays of an orthographic view: the tile lists of
    levels 6 and 7 decide by the distance across the rays."""
    used, want = check_levels(grazing_scene((0.0, 0.0, 0.0), 60, 9, mirrors=False, projection=ORTHOGRAPHIC), (0, 3, 6))
    assert used == [0, 3, 6] and (want["index"] > 0).sum() > 20
    used, _ = check_levels(grazing_scene((0.0, 0.0, 0.0), 60, 9, projection=ORTHOGRAPHIC), (0, 7))
    assert used == [0, 7]
    used, _ = check_levels(grazing_scene((1.6e4, 1.5e4, 1.7e4), 60, 10, mirrors=False, projection=ORTHOGRAPHIC), (0, 6))
    assert used == [0, 6]


def test_tangent_spheres_just_under_the_coordinate_limits(gpu):
    # |c|_1 + r just under 1e5: the ball screens still apply (the cone tables' rounded-up scale does not fit any more)
    used, _ = check_levels(grazing_scene((3.1e4, 3.0e4, 3.1e4), 60, 3), (0, 1, 2, 3))
    assert used == [0, 1, 2, 2]
    # rounded-up scale 65536 < 1e5: cone tables at their widest margins; coordinates beyond the grid's limit
    used, _ = check_levels(grazing_scene((1.6e4, 1.5e4, 1.7e4), 60, 4), (0, 2, 3))
    assert used == [0, 2, 3]
    used, _ = check_levels(grazing_scene((1.6e4, 1.5e4, 1.7e4), 60, 4, mirrors=False), (0, 6))
    assert used == [0, 6]
    used, _ = check_levels(grazing_scene((1.6e4, 1.5e4, 1.7e4), 2300, 5, size=(48, 48)), (0, 4, 5))
    assert used == [0, 4, 4]
    # just under the grid's limit of 32768
    used, _ = check_levels(grazing_scene((9.0e3, 9.5e3, 9.0e3), 2300, 6, size=(48, 48)), (0, 4, 5, 7))
    assert used == [0, 4, 5, 7]
    # negative coordinates
    used, _ = check_levels(grazing_scene((-9.0e3, 8.0e3, -9.5e3), 2300, 7, size=(48, 48)), (0, 5))
    assert used == [0, 5]


@pytest.mark.parametrize("rel", [-1e-9, 5e-10, 2e-9, 1e-6])
def test_light_and_camera_a_hair_from_a_sphere(gpu, rel):
    """A light (and the camera) at distance r (1 + rel) from a sphere's centre: around rel = 1e-9 the cone table entry of
    that sphere changes from "every line through the apex meets it" to a cone of nearly 90 degrees."""
    rt = grazing_scene((0.0, 0.0, 0.0), 80, 8)
    n = rt.numShapes
    L = rt.lights[0].loc
    r = 35.0
    rt.sphereSetup(r, [L[0] + r * (1 + rel), L[1], L[2]], n, [.8, .2, .2], 0.0, 0.1, 1.0, 0.0)
    cam = rt.camPos
    r2 = 300.0
    rt.sphereSetup(r2, [cam[0], cam[1], cam[2] + r2 * (1 + rel)], n + 1, [.2, .2, .8], 0.2, 0.1, 1.0, 0.0)
    rt.numShapes = n + 2
    check_levels(rt, (0, 1, 2, 3, 7))
    for s in rt.shapes[:rt.numShapes]:
        s.reflectivity = 0.0
    check_levels(rt, (0, 6))
