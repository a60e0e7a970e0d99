"""Soak run (not collected by pytest: run it by name on a GPU box): many small random scenes -- spheres and planes,
1 to 9 lights with and without avoid lists, mirrors, transparent shapes, depths 1 to 6, supersampling on and off --
through assert_parity: frames and index masks bit for bit, the reference's ray and test counts, and the number of
queries the library made against the oracle's prediction.
    python tests/soak_random_scenes.py [count] [first seed] [wide]
"wide": frames of up to 333 x 72 pixels (dozens of tiles per row of tiles, widths that are no multiple of the tile) and a
random ticket size, so that tile lists shared by the tiles of a ticket and tickets that straddle rows of tiles are met,
and up to 32 lights (31 is the most that candidate lists take)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import torch  # noqa: F401  (before the HIP library touches the GPU)

import raytracing_amd
from raytracing_amd import scenes, set_diagnostic
from raytracing_amd.tracer import ORTHOGRAPHIC, PERSPECTIVE
import test_gpu_parity as tp
import test_gpu_screens as ts

count = int(sys.argv[1]) if len(sys.argv) > 1 else 200
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
wide = len(sys.argv) > 3 and sys.argv[3] == "wide"
raytracing_amd.init(0)
for seed in range(first, first + count):
    rng = scenes.SplitMix64(0x50A70000 + seed)
    pick = lambda *xs: xs[int(rng.uniform(0, len(xs) - 1e-9))]  # noqa: E731
    n = pick(3, 12, 31, 32, 40, 90, 300, 1200)
    lights = pick(1, 2, 3, 5, 9, 17, 31, 32) if wide else pick(1, 2, 3, 4, 5, 7, 9)
    size = (pick(96, 200, 333, 130), pick(40, 72, 17)) if wide else (48, 40)
    if wide:
        set_diagnostic("tiles_per_ticket", pick(0, 0, 2, 3, 5, 8))
    rt = ts.scene(1000 + seed, n, size=size, lights=lights, radius=pick((2.0, 9.0), (5.0, 40.0), (20.0, 120.0)),
                  box=pick(150.0, 400.0), refl=pick(0.0, 0.4, 0.9), depth=pick(1, 2, 3, 6),
                  projection=pick(PERSPECTIVE, PERSPECTIVE, ORTHOGRAPHIC),
                  passes=pick(0, 0, 2, 3, 9) if n <= 90 and not wide else pick(0, 0, 2, 3))  # (9: no table of scale factors)
    k = rt.numShapes
    if pick(0, 1):  # two planes
        rt.planeSetup([0.0, 1.0, 0.0], [0.0, -300.0, 0.0], k, [.6, .6, .7], pick(0.0, 0.3), 0.1, 1.0, 0.0)
        rt.planeSetup([0.1, 0.0, 1.0], [0.0, 0.0, -500.0], k + 1, [.7, .5, .5], 0.0, 0.2, 1.0, 0.0)
        rt.numShapes = k = k + 2
    if pick(0, 0, 1):  # a transparent sphere
        rt.sphereSetup(60.0, [10.0, 20.0, 150.0], k, [.9, .9, 1.0], pick(0.0, 0.5), 0.1, 1.5, 0.8)
        rt.numShapes = k = k + 1
    if pick(0, 1):  # avoid lists
        for i, l in enumerate(rt.lights[:rt.numLights]):
            shapes = [rt.shapes[(7 * i + j) % rt.numShapes] for j in range(pick(0, 1, 3))]
            l.avoid, l.avoidNum = shapes, len(shapes)
    set_diagnostic("screen_cap", pick(0, 1, 2, 3, 5, 6, 7, 7))
    tp.assert_parity(rt)
    print("seed %d ok: %d shapes, %d lights, depth %d, queries made %d of %d" % (
        seed, rt.numShapes, rt.numLights, rt.maxDepth, rt.last_rays_traced, rt.last_counters[0]), flush=True)
set_diagnostic("screen_cap", 7)
set_diagnostic("tiles_per_ticket", 0)
print("soak ok: %d scenes" % count)
