"""The conservative screens (single-precision per-sphere screen, two-level ball screen, per-family cone screens)
must never change a result.  Random and adversarial sphere scenes, GPU against the CPU oracle, bit for bit, at
every screen level (the screen_cap setting, 0..7, caps the level a job may use; 0 is the plain double-precision loop).
Level 6 (candidate lists in LDS) takes scenes without reflecting shapes; the scenes of the other
tests mostly have mirrors (level 7: lists in global memory, a grid for the rays behind a mirror), so the tests at the
end of this file run their shapes again without."""
import numpy as np
import pytest

import oracle_binding as ob
from raytracing_amd import get_diagnostic, scenes, set_diagnostic
from raytracing_amd.tracer import (ANTIALIASING_OFF, ANTIALIASING_ON, GRAFT_OUT_F64_RGB, ORTHOGRAPHIC, PERSPECTIVE,
                                   Light, RayTracer, lightInit, lightSetAvoid)

pytestmark = pytest.mark.gpu


def scene(seed, n, size=(64, 64), lights=3, radius=(5.0, 40.0), box=400.0, refl=0.0, depth=4, projection=PERSPECTIVE,
          passes=0, centre=(0.0, 0.0, 100.0)):
    rng = scenes.SplitMix64(0xC0DE0000 + seed)
    rt = RayTracer(*size)
    for k in range(n):
        c = [centre[0] + rng.uniform(-box, box), centre[1] + rng.uniform(-box, box), centre[2] + rng.uniform(-box, box)]
        r = rng.uniform(*radius)
        rt.sphereSetup(r, c, k, [rng.uniform(.1, 1), rng.uniform(.1, 1), rng.uniform(.1, 1)],
                       refl if rng.uniform(0, 1) < 0.5 else 0.0, rng.uniform(0, .2), 1.0, 0.0)
    rt.numShapes = n
    for i in range(lights):
        l = Light()
        lightInit(l, [rng.uniform(.5, 1.5)] * 3, [rng.uniform(-300, 300), rng.uniform(-300, 300), rng.uniform(-200, 300)])
        rt.lights.append(l)
    rt.numLights = lights
    rt.maxDepth, rt.projectionType = depth, projection
    rt.aliasing, rt.numPasses = (ANTIALIASING_ON, passes) if passes else (ANTIALIASING_OFF, 1)
    rt.sceneInitialize([0, 0, 0], 256, 1000)
    return rt


def check_all_levels(rt, levels=(0, 1, 2, 3, 4, 5, 6, 7)):
    want = ob.render(rt, threads=16, want_index=True)
    old = get_diagnostic("screen_cap")
    try:
        for level in levels:
            set_diagnostic("screen_cap", level)
            rgb, idx = rt.render_rows(0, None, GRAFT_OUT_F64_RGB, want_index=True)
            assert np.array_equal(idx, want["index"]), "index mask differs at screen level %d" % level
            assert np.array_equal(rgb.view(np.uint64), want["rgb"].view(np.uint64)), \
                "RGB differs at screen level %d (max abs %g)" % (level, np.nanmax(np.abs(rgb - want["rgb"])))
            if level >= 6:   # launches that ask for the frame alone run kernels of their own at these levels
                plain = rt.render_rows(0, None, GRAFT_OUT_F64_RGB)
                assert np.array_equal(plain.view(np.uint64), want["rgb"].view(np.uint64)), \
                    "the frame asked for alone differs at screen level %d" % level
    finally:
        set_diagnostic("screen_cap", old)


@pytest.mark.parametrize("seed", range(4))
def test_random_scenes(gpu, seed):
    check_all_levels(scene(seed, 96 + 40 * seed, refl=0.6, depth=5))


def test_supersampled_and_orthographic(gpu):
    check_all_levels(scene(10, 150, passes=3, size=(48, 48)))
    check_all_levels(scene(11, 130, projection=ORTHOGRAPHIC))  # no camera family: primary rays use the ball table


def test_rotated_zoomed_view(gpu):
    rt = scene(12, 160, refl=0.4)
    rt.rotateView(0.4, [0.0, 1.0, 0.0])
    rt.rotateView(-0.25, list(rt.lrVec))
    rt.screenWidth *= 1.1
    rt.screenHeight *= 0.9
    check_all_levels(rt)


def test_lights_and_camera_inside_spheres(gpu):
    """Apex inside a sphere: every line through it meets that sphere, the cone entry must always pass."""
    rt = scene(13, 100)
    rt.sphereSetup(90, list(rt.lights[0].loc), 100, [.8, .3, .3], 0.0, 0.1, 1.0, 0.0)          # around light 0
    rt.sphereSetup(2000, [0, 0, -1256 + 10], 101, [.3, .8, .3], 0.3, 0.1, 1.0, 0.0)             # around the camera
    rt.sphereSetup(60, [rt.lights[1].loc[0] + 60.0, rt.lights[1].loc[1], rt.lights[1].loc[2]], 102, [.3, .3, .8], 0.0,
                   0.1, 1.0, 0.0)                                                                # light 1 ON its surface
    rt.numShapes = 103
    check_all_levels(rt)


def test_avoid_lists(gpu):
    rt = scene(14, 120, lights=3)
    lightSetAvoid(rt.lights[0], [rt.shapes[5]], 1)
    check_all_levels(rt)                       # avoid mode 1 (one shape per light)
    lightSetAvoid(rt.lights[1], [rt.shapes[7], rt.shapes[70], rt.shapes[7]], 3)
    check_all_levels(rt)                       # avoid mode 2 (general mask)


def test_duplicate_and_concentric_spheres_break_ties_by_index(gpu):
    """Equal distances: the reference keeps the lowest index.  The two-level screens visit spheres out of order."""
    rt = scene(15, 80)
    for k in range(40):
        src = rt.shapes[k]
        rt.sphereSetup(src.unif[3], src.unif[0:3], 80 + k, [.9, .1, .1], 0.0, 0.1, 1.0, 0.0)   # exact duplicates
    rt.numShapes = 120
    check_all_levels(rt)
    idx = rt.render_rows(0, None, GRAFT_OUT_F64_RGB, want_index=True)[1]
    assert idx.max() < 80 and (idx >= 0).any()  # a duplicate never wins over its lower-indexed original


def test_nearly_coincident_spheres(gpu):
    """Hit distances that differ by a few units in the last place: the walks compare ray parameters first and must
    fall back to the reference's comparison of distances exactly when those cannot decide."""
    rt = scene(22, 80, refl=0.5, depth=3)
    for k in range(40):
        src = rt.shapes[k]
        eps = [0.0, 1e-15, 3e-14, 1e-13, 1e-12, 1e-10, 1e-8][k % 7]
        c = [src.unif[0] * (1 + eps), src.unif[1] * (1 - eps), src.unif[2] + eps * 100]
        rt.sphereSetup(src.unif[3] * (1 + (eps if k % 2 else -eps)), c, 80 + k, [.9, .1, .1], 0.3, 0.1, 1.0, 0.0)
    rt.numShapes = 120
    check_all_levels(rt)
    rt = large_scene(34, 1300, lights=5, refl=0.5, depth=3)
    n = rt.numShapes
    for k in range(30):
        src = rt.shapes[11 * k]
        eps = [1e-15, 1e-13, 1e-11][k % 3]
        rt.sphereSetup(src.unif[3] * (1 + eps), [src.unif[0] + eps * 50, src.unif[1], src.unif[2] * (1 - eps)], n + k,
                       [.2, .9, .2], 0.3, 0.1, 1.0, 0.0)
    rt.numShapes = n + 30
    check_all_levels(rt, levels=(0, 4, 5, 7))


def test_tiny_giant_and_far_spheres(gpu):
    rt = scene(16, 90, radius=(0.01, 3.0))
    rt.sphereSetup(5000, [0, 0, 200], 90, [.5, .5, .9], 0.2, 0.1, 1.0, 0.0)      # encloses everything
    rt.sphereSetup(1e-3, [10, 10, 50], 91, [1, 1, 1], 0.0, 0.1, 1.0, 0.0)
    rt.numShapes = 92
    check_all_levels(rt)


def test_vanishing_radii(gpu):
    """Squared radii in and below the denormal range (and zero): the exact loop's fused miss test is switched off for
    such a job; a camera inside a sphere and a sphere centred on a light keep the test's "inside" side busy."""
    rt = scene(41, 60)
    rt.sphereSetup(1e-150, [5, -3, 60], 60, [1, 1, 1], 0.0, 0.1, 1.0, 0.0)
    rt.sphereSetup(0.0, [-20, 12, 80], 61, [1, 1, 1], 0.0, 0.1, 1.0, 0.0)
    rt.sphereSetup(3e-162, [30, 30, 30], 62, [1, 1, 1], 0.3, 0.1, 1.0, 0.0)
    rt.numShapes = 63
    check_all_levels(rt, levels=(0, 1, 3))
    rt = scene(42, 60)
    rt.camInitialize(rt.target, rt.targetToScreen, rt.screenToCam)
    cam, L = rt.camPos, rt.lights[0].loc
    rt.sphereSetup(2000.0, [cam[0] + 3, cam[1] - 2, cam[2] + 100], 60, [.4, .6, .9], 0.2, 0.1, 1.0, 0.0)
    rt.sphereSetup(25.0, [L[0], L[1], L[2]], 61, [.9, .9, .2], 0.0, 0.1, 1.0, 0.0)
    rt.numShapes = 62
    check_all_levels(rt, levels=(0, 1, 3))


def test_large_coordinates_fall_back(gpu):
    """Coordinates beyond 1e5 disable the two-level screens, beyond 1e15 the per-sphere screen; results stay exact."""
    rt = scene(17, 100, centre=(3e5, -2e5, 1e5), box=300.0)
    rt.sceneInitialize([3e5, -2e5, 1e5], 256, 1000)
    for l in rt.lights[:rt.numLights]:
        l.loc = [l.loc[0] + 3e5, l.loc[1] - 2e5, l.loc[2] + 1e5]
    check_all_levels(rt)
    rt = scene(18, 70)
    rt.sphereSetup(1e15, [0, 0, 3e16], 70, [.2, .9, .4], 0.0, 0.1, 1.0, 0.0)
    rt.numShapes = 71
    check_all_levels(rt)


def test_many_lights(gpu):
    check_all_levels(scene(19, 100, lights=7))
    check_all_levels(scene(20, 100, lights=9))  # more families than the cone tables support: ball screen, or lists
    check_all_levels(scene(22, 140, lights=31, size=(40, 40)), levels=(0, 2, 6))
    check_all_levels(scene(23, 140, lights=19, size=(40, 40), refl=0.5, depth=3), levels=(0, 7))


def test_transparent_with_many_spheres(gpu):
    rt = scene(21, 90, refl=0.5, depth=5)
    for k in (3, 17, 40, 41):
        rt.shapes[k].transparency = 0.7
        rt.shapes[k].refrIndex = 1.3 + 0.01 * k
    check_all_levels(rt)


def large_scene(seed, n, **kw):
    """More spheres than the per-family tables can keep in LDS: the three-level walk (level 4)."""
    rt = scene(seed, n, size=(48, 48), radius=(3.0, 14.0), **kw)
    rt.sphereSetup(70, list(rt.lights[0].loc), n, [.8, .3, .3], 0.0, 0.1, 1.0, 0.0)              # around light 0
    rt.sphereSetup(1500, [0, 0, -1256 + 10], n + 1, [.3, .8, .3], 0.3, 0.1, 1.0, 0.0)            # around the camera
    for k in range(30):                                                                          # exact duplicates
        src = rt.shapes[7 * k]
        rt.sphereSetup(src.unif[3], src.unif[0:3], n + 2 + k, [.9, .1, .1], 0.0, 0.1, 1.0, 0.0)
    rt.numShapes = n + 32
    return rt


def test_large_scenes_three_level_walk(gpu):
    check_all_levels(large_scene(30, 1500, lights=5, refl=0.5, depth=3), levels=(0, 2, 4, 5, 7))
    check_all_levels(large_scene(31, 1237, projection=ORTHOGRAPHIC, lights=7, refl=0.5, depth=3), levels=(0, 4, 5))
    rt = large_scene(32, 2100, lights=4)
    lightSetAvoid(rt.lights[2], [rt.shapes[11]], 1)
    check_all_levels(rt, levels=(0, 4, 5, 7))
    lightSetAvoid(rt.lights[0], [rt.shapes[3], rt.shapes[2101]], 2)   # (2101: the sphere around the camera)
    check_all_levels(rt, levels=(0, 7))
    check_all_levels(matte(large_scene(36, 2100, lights=4)), levels=(0, 5, 7))


def test_grid_walk_far_from_the_scene(gpu):
    """An orthographic view whose screen plane lies tens of thousands of units from a compact scene: every primary
    ray enters the grid at a ray parameter of about 3e4, where single precision resolves 2e-3 -- the walk must step
    relative to its entry point, or cells that list the sphere hit are skipped."""
    rt = large_scene(35, 1500, projection=ORTHOGRAPHIC, lights=7, refl=0.5, depth=3)
    rt.sceneInitialize([0.0, 0.0, 0.0], 30000.0, 1000.0)
    check_all_levels(rt, levels=(0, 5))
    rt.rotateView(0.7, [0.3, 1.0, 0.1])
    rt.rotateView(-0.4, list(rt.lrVec))
    check_all_levels(rt, levels=(0, 5))


def test_groups_of_sixteen_and_other_sizes(gpu):
    """Scenes of 512+ spheres whose tables fit in LDS use 16 spheres per group; GRAFT_GROUP forces other sizes."""
    rt = large_scene(33, 600, lights=3, refl=0.4, depth=3)
    check_all_levels(rt, levels=(0, 2, 3))
    old = get_diagnostic("group_size")
    try:
        for size in (4, 12, 32):
            set_diagnostic("group_size", size)
            check_all_levels(rt, levels=(2, 3))
            check_all_levels(large_scene(30, 1500, lights=5, refl=0.5, depth=3), levels=(4, 5))
    finally:
        set_diagnostic("group_size", old)


def test_levels_are_actually_used(gpu):
    """The scenes above must really reach the screens they are meant to test."""
    def level(rt, cap=7):
        old = set_diagnostic("screen_cap", cap)
        try:
            job = rt.create_job()
            lv = job.screen_level()
            job.close()
            return lv
        finally:
            set_diagnostic("screen_cap", old)
    assert level(scene(0, 96)) == 6 and level(scene(0, 96, refl=0.5)) == 7 and level(scene(0, 96, refl=0.5), 6) == 3
    assert [level(scene(0, 96), cap) for cap in (0, 1, 2, 3, 5)] == [0, 1, 2, 3, 3]
    assert level(scenes.workload("cfg4", 64, 64)) == 6 and level(scenes.workload("cfg3", 64, 64)) == 6
    assert level(scenes.workload("cfg4", 64, 64), 5) == 3 and level(scenes.workload("cfg3", 64, 64), 5) == 3
    assert level(scenes.workload("cfg5", 64, 64)) == 7 and level(scenes.workload("cfg5", 64, 64), 4) == 4
    assert level(scenes.workload("cfg5", 64, 64), 6) == 5
    assert level(large_scene(30, 1500, lights=5)) == 7 and level(large_scene(30, 1500, lights=5), 3) == 2
    assert level(large_scene(30, 1500, lights=5), 5) == 5
    assert level(matte(large_scene(32, 2100, lights=4))) == 7   # its list tables do not fit in LDS
    assert level(matte(large_scene(32, 2100, lights=4)), 6) == 5
    assert level(matte(large_scene(33, 600, lights=3))) == 6
    assert level(scene(11, 130, projection=ORTHOGRAPHIC)) == 6    # tile lists of parallel rays
    assert level(scene(11, 130, projection=ORTHOGRAPHIC), 5) == 3
    assert level(large_scene(31, 1237, projection=ORTHOGRAPHIC, lights=7)) == 7
    assert level(large_scene(31, 1237, projection=ORTHOGRAPHIC, lights=7), 5) == 5
    far = large_scene(35, 1500, projection=ORTHOGRAPHIC, lights=7, refl=0.5, depth=3)
    far.sceneInitialize([0.0, 0.0, 0.0], 30000.0, 1000.0)
    assert level(far) == 7 and level(far, 5) == 5
    assert level(scene(20, 100, lights=9)) == 6 and level(scene(20, 100, lights=9), 5) == 2   # too many lights for the family tables, not for the lists
    assert level(scene(20, 100, lights=32)) == 2
    assert level(scene(1, 40)) == 1                      # too few spheres for grouping
    assert level(scenes.default_scene()) == 1            # planes: per-sphere screen only
    rt = scene(21, 90)
    rt.shapes[3].transparency = 0.5
    assert level(rt) == 1                                # transparent: tree-walk kernels, per-sphere screen



# ---- level 6: candidate lists (scenes without reflecting shapes) -------------------------------------

def matte(rt):
    """The same scene with every reflectivity set to 0."""
    for s in rt.shapes[:rt.numShapes]:
        s.reflectivity = 0.0
    return rt


@pytest.mark.parametrize("seed", range(4))
def test_lists_random_scenes(gpu, seed):
    check_all_levels(scene(50 + seed, 96 + 40 * seed, lights=1 + seed), levels=(0, 3, 6))


def test_lists_supersampled_rotated_zoomed(gpu):
    check_all_levels(scene(54, 150, passes=3, size=(48, 48)), levels=(0, 6))
    rt = scene(55, 160, size=(72, 40))
    rt.rotateView(0.4, [0.0, 1.0, 0.0])
    rt.rotateView(-0.25, list(rt.lrVec))
    rt.screenWidth *= 1.1
    rt.screenHeight *= 0.9
    check_all_levels(rt, levels=(0, 6))
    rt.aliasing, rt.numPasses = ANTIALIASING_ON, 2
    check_all_levels(rt, levels=(0, 6))


def test_lists_shared_by_the_tiles_of_a_ticket(gpu):
    """Launches of fewer than four samples per pixel draw tickets of several 8x8 tiles, and the tiles of a ticket that lie
    in one row of tiles share one tile list (built for the rectangle they cover).  Frames whose width is no multiple of
    the tile, with tickets of 3, 5 and 16 tiles forced -- tickets that straddle rows of tiles, tickets longer than a
    row --, one sample per pixel and supersampling with one pass; with mirrors (level 7, a list per tile) as well."""
    old = get_diagnostic("tiles_per_ticket")
    try:
        for per_ticket in (3, 5, 16):
            set_diagnostic("tiles_per_ticket", per_ticket)
            rt = scene(70 + per_ticket, 120, size=(100, 44), radius=(4.0, 25.0))
            check_all_levels(rt, levels=(0, 7))          # mirrors: level 7
            check_all_levels(matte(rt), levels=(0, 6, 7))
            rt.aliasing, rt.numPasses = ANTIALIASING_ON, 1
            check_all_levels(rt, levels=(0, 6))
            rt.aliasing, rt.numPasses = ANTIALIASING_ON, 2   # four samples: a list per tile, whatever the ticket
            check_all_levels(rt, levels=(0, 6))
        set_diagnostic("tiles_per_ticket", 4)
        rt = matte(scene(79, 200, size=(13, 90), radius=(3.0, 12.0)))   # two tiles per row: every ticket spans rows
        check_all_levels(rt, levels=(0, 6, 7))
    finally:
        set_diagnostic("tiles_per_ticket", old)


def test_lists_orthographic_views(gpu):
    """Orthographic views (mainTracing.c:410-413: every primary ray has the direction -viewDir): the tile list holds the
    spheres whose distance from the tile's centre line, measured across the rays, is within their radius plus the tile's
    reach.  Plain, rotated and zoomed, supersampled, with mirrors (level 7), seen from tens of thousands of units away,
    and with view vectors that are NOT perpendicular to one another (the reference never produces such a view; the
    library takes what it is given)."""
    check_all_levels(scene(80, 150, projection=ORTHOGRAPHIC), levels=(0, 6, 7))
    rt = scene(81, 180, projection=ORTHOGRAPHIC, size=(72, 40), radius=(4.0, 30.0))
    rt.rotateView(0.5, [0.2, 1.0, 0.0])
    rt.rotateView(-0.3, list(rt.lrVec))
    rt.screenWidth *= 0.9
    rt.screenHeight *= 1.1
    check_all_levels(rt, levels=(0, 6))
    rt.aliasing, rt.numPasses = ANTIALIASING_ON, 3
    check_all_levels(rt, levels=(0, 6))
    rt.aliasing, rt.numPasses = ANTIALIASING_ON, 1
    check_all_levels(rt, levels=(0, 6))
    check_all_levels(scene(82, 200, projection=ORTHOGRAPHIC, refl=0.6, depth=4), levels=(0, 7))
    far = large_scene(83, 1500, projection=ORTHOGRAPHIC, lights=5, refl=0.5, depth=3)
    far.sceneInitialize([0.0, 0.0, 0.0], 30000.0, 1000.0)
    far.rotateView(0.7, [0.3, 1.0, 0.1])
    check_all_levels(far, levels=(0, 7))
    skew = scene(84, 160, projection=ORTHOGRAPHIC, size=(56, 48))
    skew.lrVec = [1.0, 0.15, -0.3]     # neither unit length nor perpendicular to viewDir or udVec
    skew.udVec = [0.1, 0.9, 0.2]
    check_all_levels(skew, levels=(0, 6))


def test_lists_lights_and_camera_inside_or_on_spheres(gpu):
    """The camera inside a sphere: that sphere is on every tile's list.  A light inside or on a sphere: that sphere is
    on every occluder list of the light, and a shadow ray that starts on it may meet any sphere."""
    rt = scene(56, 100)
    rt.sphereSetup(90, list(rt.lights[0].loc), 100, [.8, .3, .3], 0.0, 0.1, 1.0, 0.0)          # around light 0
    rt.sphereSetup(2000, [0, 0, -1256 + 10], 101, [.3, .8, .3], 0.0, 0.1, 1.0, 0.0)             # around the camera
    rt.sphereSetup(60, [rt.lights[1].loc[0] + 60.0, rt.lights[1].loc[1], rt.lights[1].loc[2]], 102, [.3, .3, .8], 0.0,
                   0.1, 1.0, 0.0)                                                                # light 1 ON its surface
    rt.sphereSetup(30, [rt.lights[2].loc[0] + 30.0 * (1 + 1e-7), rt.lights[2].loc[1], rt.lights[2].loc[2]], 103,
                   [.9, .9, .3], 0.0, 0.1, 1.0, 0.0)                                             # light 2 a hair outside
    rt.numShapes = 104
    check_all_levels(rt, levels=(0, 3, 6))


def test_lists_avoid_lists(gpu):
    rt = scene(57, 120, lights=3)
    lightSetAvoid(rt.lights[0], [rt.shapes[5]], 1)
    check_all_levels(rt, levels=(0, 6))
    lightSetAvoid(rt.lights[1], [rt.shapes[7], rt.shapes[70], rt.shapes[7]], 3)
    check_all_levels(rt, levels=(0, 6))


def test_lists_duplicates_and_near_duplicates(gpu):
    rt = scene(58, 80)
    for k in range(40):
        src = rt.shapes[k]
        eps = [0.0, 0.0, 1e-15, 3e-14, 1e-12, 1e-10, 1e-8][k % 7]
        c = [src.unif[0] * (1 + eps), src.unif[1] * (1 - eps), src.unif[2] + eps * 100]
        rt.sphereSetup(src.unif[3] * (1 + (eps if k % 2 else -eps)), c, 80 + k, [.9, .1, .1], 0.0, 0.1, 1.0, 0.0)
    rt.numShapes = 120
    check_all_levels(rt, levels=(0, 6))


def test_lists_tiny_giant_and_overlapping_spheres(gpu):
    rt = scene(59, 90, radius=(0.01, 3.0))
    rt.sphereSetup(5000, [0, 0, 200], 90, [.5, .5, .9], 0.0, 0.1, 1.0, 0.0)      # encloses everything, lights included
    rt.sphereSetup(1e-3, [10, 10, 50], 91, [1, 1, 1], 0.0, 0.1, 1.0, 0.0)
    rt.numShapes = 92
    check_all_levels(rt, levels=(0, 6))
    rt = scene(60, 140, radius=(60.0, 160.0), box=300.0)                         # every sphere cuts through several others
    check_all_levels(rt, levels=(0, 3, 6))
    rt = scene(61, 70)
    rt.sphereSetup(1e-150, [5, -3, 60], 70, [1, 1, 1], 0.0, 0.1, 1.0, 0.0)
    rt.sphereSetup(0.0, [-20, 12, 80], 71, [1, 1, 1], 0.0, 0.1, 1.0, 0.0)
    rt.numShapes = 72
    check_all_levels(rt, levels=(0, 6))


def test_lists_more_survivors_than_one_round_holds(gpu):
    """The list walk keeps up to four spheres that passed the single-precision screen before a round of exact tests;
    rays through a row of fourteen spheres lined up behind one another along the view (and, seen from a light placed
    on that line, along the shadow rays) pass the screen of all of them: several rounds per ray, lanes whose lists end
    early beside lanes with a dozen survivors.  Nested and equal spheres among them (ties go to the lower index)."""
    rt = matte(scene(65, 60, size=(56, 48), lights=2, radius=(3.0, 10.0)))
    k = rt.numShapes
    for i in range(14):
        r = [30.0, 30.0, 18.0, 40.0, 30.0][i % 5]
        rt.sphereSetup(r, [4.0 * (i % 3), -3.0 * (i % 2), 40.0 + 35.0 * i], k + i, [.2 + .05 * i, .5, .9 - .05 * i],
                       0.0, 0.1, 1.0, 0.0)
    rt.numShapes = k + 14
    rt.lights[1].loc = [3.0, -1.0, -200.0]          # on the line of the row: its shadow rays run along it
    check_all_levels(rt, levels=(0, 6, 7))
    rt.aliasing, rt.numPasses = ANTIALIASING_ON, 2
    check_all_levels(rt, levels=(0, 6))


def test_lists_supersampled_without_a_table_of_scale_factors(gpu):
    """Supersampled frames whose table of sample scale factors is not in LDS -- more than 8 passes per pixel side -- run
    the list kernels that work the two getNewPoint steps out per sample (mainTracing.c:405-408)."""
    rt = matte(scene(66, 100, size=(24, 16), lights=2, radius=(4.0, 25.0)))
    rt.aliasing, rt.numPasses = ANTIALIASING_ON, 9
    check_all_levels(rt, levels=(0, 6, 7))
    rt = scene(67, 100, size=(16, 16), lights=2, radius=(4.0, 25.0), refl=0.5, depth=3)
    rt.aliasing, rt.numPasses = ANTIALIASING_ON, 10
    check_all_levels(rt, levels=(0, 7))


def test_lists_tile_list_overflow_and_wide_views(gpu):
    """More spheres in a tile's cone than its list holds (a small frame of many spheres): the tile falls back to the
    list of all spheres; and a view so wide that a tile has no usable bounding cone."""
    rt2 = scene(62, 700, size=(16, 16), radius=(3.0, 14.0), lights=2)
    check_all_levels(rt2, levels=(0, 6))
    rt3 = scene(63, 100, size=(24, 24))
    rt3.screenWidth = rt3.screenHeight = 40000.0     # 8-pixel tiles span tens of degrees
    check_all_levels(rt3, levels=(0, 6))
    rt4 = scene(64, 100, size=(24, 24))
    rt4.screenWidth = rt4.screenHeight = 3.0         # zoomed far in: most tiles see nothing or one sphere
    check_all_levels(rt4, levels=(0, 6))


def test_lists_seven_lights_and_large_coordinates(gpu):
    check_all_levels(scene(65, 100, lights=7), levels=(0, 6))
    rt = scene(66, 100, centre=(1.6e4, 1.5e4, 1.7e4), box=300.0)
    rt.sceneInitialize([1.6e4, 1.5e4, 1.7e4], 256, 1000)
    for l in rt.lights[:rt.numLights]:
        l.loc = [l.loc[0] + 1.6e4, l.loc[1] + 1.5e4, l.loc[2] + 1.7e4]
    check_all_levels(rt, levels=(0, 3, 6))


def test_lists_six_hundred_spheres(gpu):
    check_all_levels(matte(large_scene(33, 600, lights=3)), levels=(0, 3, 6))
    old = get_diagnostic("group_size")
    try:
        for size in (4, 32):
            set_diagnostic("group_size", size)
            check_all_levels(matte(large_scene(33, 600, lights=3)), levels=(6,))
    finally:
        set_diagnostic("group_size", old)
