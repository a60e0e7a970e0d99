"""Host-side steps of launchRays in the product (C ABI helpers + the Python mirror) against the oracle's own
restatement and the golden state recorded from the real reference.  No GPU needed."""
import numpy as np
import pytest

import oracle_binding as ob
from cases import CASES, build
from golden_util import GOLDEN
from raytracing_amd.distributed import band_rows
from raytracing_amd._lib import lib


@pytest.mark.parametrize("extent,n", [(512.0, 512), (512.0 * .9, 512), (512.0 * 1.1, 512), (512 * 1.1 * 1.1 * .9, 64),
                                      (512.0, 40), (512.0, 4096), (512.0, 16384), (460.8 * .9 * .9, 96), (3.7, 7)])
def test_loop_offsets(extent, n):
    """Accumulated-double loop counters of mainTracing.c:392-393, including the 513th iteration after zooming."""
    want = ob.loop_offsets(extent, n)
    buf = np.zeros(2 * n + 32)
    count = lib.graftLoopOffsets(extent, n, buf.ctypes.data_as(ob._dp), len(buf))
    assert count == len(want)
    assert np.array_equal(buf[:count].view(np.uint64), want.view(np.uint64))
    for c in want[:: max(1, len(want) // 50)]:
        assert lib.graftScreenCoord(float(c), n, extent) == ob.screen_coord(c, n, extent)


def test_loop_offsets_rejects_loops_that_never_end():
    for extent in (0.0, -5.0, float("nan"), float("inf")):
        assert lib.graftLoopOffsets(extent, 16, None, 0) == -1


def test_zoom_out_runs_513_iterations():
    assert len(ob.loop_offsets(512.0 * 1.1, 512)) == 513
    assert GOLDEN["default_zoom_out"]["calls"] == 513 * 513 and GOLDEN["default_zoom_out"]["out_of_range"] == 1025


@pytest.mark.parametrize("name", ["default_aa6", "default_rot_left", "default_zoom_out", "keys_05", "keys_09",
                                  "keys_15", "mixed_6_rotated"])
def test_host_state_matches_reference_globals(name):
    """After the same operations the mirror's globals equal the real reference's (recorded in golden.json),
    bit for bit: rotateView (sin/cos, mat33AngleAxisRotation), zoom, flag toggles, camInitialize."""
    rt = build(name)
    rt.camInitialize(rt.target, rt.targetToScreen, rt.screenToCam)
    g = GOLDEN[name]["state"]
    for key in ("viewDir", "lrVec", "udVec", "screenCenter", "camPos", "target"):
        assert np.array_equal(np.array(getattr(rt, key)).view(np.uint64), np.array(g[key]).view(np.uint64)), key
    for key in ("screenWidth", "screenHeight", "projectionType", "aliasing", "numPasses", "maxDepth", "numShapes"):
        assert getattr(rt, key) == g[key], key
    sc, cp = ob.cam_initialize(rt.target, rt.viewDir, rt.targetToScreen, rt.screenToCam)
    assert np.array_equal(sc, np.array(rt.screenCenter)) and np.array_equal(cp, np.array(rt.camPos))


def test_pack_translates_pointers_to_indices():
    rt = build("mixed_2")
    scene, view, keep = rt.pack()
    assert scene.numShapes == 15 and scene.numLights == 3
    lights = [scene.lights[i] for i in range(3)]
    assert [l.avoidNum for l in lights] == [0, 1, 3]
    assert [scene.avoidIndex[i] for i in range(scene.avoidIndexLen)] == [1, 0, 14, 3]
    assert [scene.shapes[k].kind for k in range(15)] == [0] * 12 + [1] * 3
    assert view.nCols == 64 and view.nRows == 64 and view.aliasing == 0


@pytest.mark.parametrize("rows,world", [(16384, 8), (4096, 4), (512, 3), (100, 8), (8, 8), (1, 2), (72, 5)])
def test_band_rows_partition(rows, world):
    bands = [band_rows(rows, world, r) for r in range(world)]
    assert bands[0][0] == 0 and sum(n for _, n in bands) == rows
    for (a, n), (b, _) in zip(bands, bands[1:]):
        assert a + n == b and (n == 0 or a % 8 == 0)


def test_write_ppm(tmp_path):
    """Headless output: P6, top row first, the 8-bit quantiser of the RGB bar."""
    from raytracing_amd import GRAFT_OUT_F32_RGBA, GRAFT_OUT_F64_RGB, GRAFT_OUT_U8_RGBA, write_ppm
    rng = np.random.default_rng(3)
    f64 = rng.uniform(-0.2, 1.3, (5, 7, 3))
    f64[0, 0] = [np.nan, 0.5, 1.0]
    want = (np.clip(np.nan_to_num(f64), 0, 1) * 255 + 0.5).astype(np.uint8)[::-1]
    for fmt, frame in ((GRAFT_OUT_F64_RGB, f64),
                       (GRAFT_OUT_F32_RGBA, np.concatenate([f64, np.ones((5, 7, 1))], 2).astype(np.float32)),
                       (GRAFT_OUT_U8_RGBA, np.concatenate([want[::-1], np.full((5, 7, 1), 255, np.uint8)], 2))):
        path = tmp_path / ("f%d.ppm" % fmt)
        write_ppm(path, frame, fmt)
        raw = open(path, "rb").read()
        assert raw.startswith(b"P6\n7 5\n255\n")
        got = np.frombuffer(raw[len(b"P6\n7 5\n255\n"):], dtype=np.uint8).reshape(5, 7, 3)
        if fmt == GRAFT_OUT_F32_RGBA:
            want32 = (np.clip(np.nan_to_num(frame[..., :3].astype(np.float64)), 0, 1) * 255 + 0.5).astype(np.uint8)[::-1]
            assert np.array_equal(got, want32)
        else:
            assert np.array_equal(got, want)


def test_write_png(tmp_path):
    """Headless output as PNG: decoded by an independent reader (PIL) it is the PPM's picture; a frame whose scanlines
    span several stored deflate blocks included."""
    from PIL import Image
    from raytracing_amd import GRAFT_OUT_F32_RGBA, GRAFT_OUT_F64_RGB, write_png
    rng = np.random.default_rng(5)
    for shape in ((5, 7), (1, 1), (300, 211)):   # 300 x 211: 190 200 bytes of scanlines = three stored blocks
        f64 = rng.uniform(-0.2, 1.3, shape + (3,))
        f64[0, 0] = [np.nan, 0.5, 1.0]
        want = (np.clip(np.nan_to_num(f64), 0, 1) * 255 + 0.5).astype(np.uint8)[::-1]
        path = tmp_path / "f.png"
        write_png(path, f64, GRAFT_OUT_F64_RGB)
        with Image.open(path) as im:
            assert im.mode == "RGB" and im.size == (shape[1], shape[0])
            assert np.array_equal(np.asarray(im), want)
        f32 = np.concatenate([f64, np.ones(shape + (1,))], 2).astype(np.float32)
        write_png(path, f32, GRAFT_OUT_F32_RGBA)
        with Image.open(path) as im:
            want32 = (np.clip(np.nan_to_num(f32[..., :3].astype(np.float64)), 0, 1) * 255 + 0.5).astype(np.uint8)[::-1]
            assert np.array_equal(np.asarray(im), want32)
    with pytest.raises(Exception):
        write_png(tmp_path / "no_such_dir" / "f.png", f64, GRAFT_OUT_F64_RGB)


@pytest.mark.parametrize("rows,world,chunks", [(16384, 8, 8), (8192, 4, 11), (4096, 2, 16), (512, 3, 5), (100, 8, 8),
                                               (8, 8, 4), (1, 2, 3), (72, 5, 1), (4097, 7, 9)])
def test_c_piece_list_equals_the_python_one(rows, world, chunks):
    """graftFramePieces (what graftRenderFrameMulti deals to the GPUs of one process) against
    raytracing_amd.distributed.frame_pieces (what the one-process-per-GPU path deals to the ranks)."""
    from raytracing_amd import frame_pieces as c_pieces
    from raytracing_amd.distributed import frame_pieces
    got = c_pieces(rows, world, chunks)
    assert got == frame_pieces(rows, world, chunks)
    assert got[0][0] == 0 and sum(n for _, n, _ in got) == rows
    for (a, n, _), (b, _, _) in zip(got, got[1:]):
        assert a + n == b and a % 8 == 0


@pytest.mark.parametrize("rows", [16384, 8192, 4097, 513, 100, 64, 9, 8, 1])
@pytest.mark.parametrize("devices", [2, 3, 4, 5, 6, 7, 8])
@pytest.mark.parametrize("per_device", [1, 3, 8, 16])
def test_c_frame_plan(rows, devices, per_device):
    """graftFramePlan (where graftRenderFrameMulti puts every piece): pieces cover every row once, in order; a device's
    pieces lie back to back in its own buffer, in the order they are traced; the buffer sizes are the sums; and the
    whole plan equals what every rank of the one-process-per-GPU path derives for itself (owned_pieces)."""
    from raytracing_amd import frame_pieces as c_pieces, frame_plan
    from raytracing_amd.distributed import local_rows, owned_pieces
    pieces, owned = frame_plan(rows, devices, per_device)
    assert [(a, n, o) for a, n, o, _ in pieces] == c_pieces(rows, devices, per_device)
    assert pieces[0][0] == 0 and sum(n for _, n, _, _ in pieces) == rows and len(owned) == devices
    covered = np.zeros(rows, dtype=int)
    for first, n, owner, at in pieces:
        assert n > 0 and 0 <= owner < devices and first % 8 == 0
        covered[first:first + n] += 1
    assert (covered == 1).all()
    for d in range(devices):
        mine = [(first, n, at) for first, n, owner, at in pieces if owner == d]
        assert mine == owned_pieces(rows, devices, d, per_device)
        at = 0
        for _, n, where in mine:          # disjoint, gap-free, in tracing order
            assert where == at
            at += n
        assert owned[d] == at == local_rows(rows, devices, d, per_device)
    # with an index plane the same row offsets address 4-byte entries: nothing else to plan
