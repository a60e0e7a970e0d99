"""Helpers shared by the golden-vector tests: load tests/golden/, replay the pixSetRGB mapping, compare."""
import json
import os

import numpy as np

import oracle_binding as ob

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

with open(os.path.join(GOLDEN_DIR, "golden.json")) as _f:
    GOLDEN = json.load(_f)["cases"]
FRAMES = np.load(os.path.join(GOLDEN_DIR, "frames.npz"))


def display(st, rgb, index, cols, rows):
    """What the reference's display holds after launchRays: frame[y][x] and mask[y][x] from loop-indexed
    arrays, replaying pixSetRGB(getScreenCoordX(i), getScreenCoordY(j), ...) in call order (x outer, y inner,
    mainTracing.c:392-393, :439) with the ORACLE's restatement of getScreenCoordX/Y.  Returns
    (frame, mask, call log as int32 [calls][2], calls outside the window)."""
    W, H = int(st.WIDTH), int(st.HEIGHT)
    px = np.array([ob.screen_coord(i, W, st.screenWidth) for i in cols], dtype=np.int64)
    py = np.array([ob.screen_coord(j, H, st.screenHeight) for j in rows], dtype=np.int64)
    xs = np.repeat(px, len(py))
    ys = np.tile(py, len(px))
    log = np.stack([xs, ys], axis=1).astype(np.int32)
    ok = (xs >= 0) & (xs < W) & (ys >= 0) & (ys < H)
    frame = np.empty((H, W, 3))
    frame[:] = np.asarray(st.backgroundColor, dtype=np.float64)
    mask = np.full((H, W), -7, dtype=np.int32)
    ia = np.repeat(np.arange(len(px)), len(py))[ok]
    ib = np.tile(np.arange(len(py)), len(px))[ok]
    flat = (ys[ok] * W + xs[ok])
    # later calls overwrite earlier ones: keep the last call per pixel
    _, first_from_end = np.unique(flat[::-1], return_index=True)
    last = len(flat) - 1 - first_from_end
    if rgb is not None:
        frame.reshape(-1, 3)[flat[last]] = rgb[ib[last], ia[last]]
    if index is not None:
        mask.reshape(-1)[flat[last]] = index[ib[last], ia[last]]
    return frame, mask, log, int(np.count_nonzero(~ok))


def check_against_golden(name, st, rgb, index, cols, rows, check_rgb_bits=True):
    g = GOLDEN[name]
    frame, mask, log, oob = display(st, rgb, index, cols, rows)
    assert [st.WIDTH, st.HEIGHT] == g["size"]
    assert len(log) == g["calls"], "pixSetRGB call count"
    assert oob == g["out_of_range"], "calls outside the window"
    assert ob.fnv1a64(log) == g["log_fnv"], "pixSetRGB (x, y) call sequence"
    small = st.WIDTH * st.HEIGHT <= 96 * 72
    want_rgb, want_idx = FRAMES[name + "/rgb"], FRAMES[name + "/idx"]
    got_rgb = frame if small else frame[::16, ::16]
    got_idx = mask if small else mask[::16, ::16]
    if index is not None:
        assert np.array_equal(got_idx, want_idx), "index mask differs from the stored reference pixels"
        assert ob.fnv1a64(mask) == g["idx_fnv"], "index mask fingerprint"
    if rgb is not None:
        q = lambda v: (np.clip(np.nan_to_num(v), 0.0, 1.0) * 255 + 0.5).astype(np.int64)
        assert np.abs(q(got_rgb) - q(want_rgb)).max() <= 1, "RGB off by more than 1 LSB of 8 bit"
        if check_rgb_bits:
            assert np.array_equal(got_rgb.view(np.uint64), want_rgb.view(np.uint64)), \
                "RGB differs from the stored reference pixels (max abs %g)" % np.nanmax(np.abs(got_rgb - want_rgb))
            assert ob.fnv1a64(frame) == g["rgb_fnv"], "frame fingerprint"
    return frame, mask
