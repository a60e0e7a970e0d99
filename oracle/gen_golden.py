#!/usr/bin/env python3
"""Generate tests/golden/ from the REAL reference (oracle/_ref, built in place from /root/reference).

    make -C oracle ref && python oracle/gen_golden.py

Every case of tests/cases.py is interpreted against the reference's own functions and globals (through the
setters of oracle/ref_harness_body.c), one frame is traced with the reference's launchRays(), and what its
pixSetRGB calls delivered is recorded:
  golden.json   per case: FNV-1a64 of the displayed frame (double[H][W][3]) and of the primary closest-shape
                index mask (int32[H][W]), number of pixSetRGB calls, how many fell outside the window, FNV of
                the (x, y) call sequence, min/max, index histogram, a few probe pixels
  frames.npz    per case: the whole frame and mask for frames up to 96x72, a [::16, ::16] subsample otherwise
The fixtures are data only; no reference source is stored.  Unwritten pixels hold the background colour
(handleKeyDown clears to it before tracing, mainTracing.c:629) and -7 in the mask.
"""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_binding as ob  # noqa: E402  (only for the FNV helper)
import ref_binding as rb  # noqa: E402
from cases import CASES  # noqa: E402

_refs = {}


def reference_for(size):
    key = tuple(size) if size else None
    if key not in _refs:
        _refs[key] = rb.Reference(key)
    return _refs[key]


def run_case(name):
    size, ops = CASES[name]
    R = reference_for(size)
    R.clear_scene()
    R.set_screen(512.0, 512.0)
    R.set_background((0, 0, 0))
    R.set_flags(rb.PERSPECTIVE, 0, 1, 4)
    bg = (0.0, 0.0, 0.0)
    for op in ops:
        kind = op[0]
        if kind == "default":
            R.default_scene()
            bg = (0.0, 0.0, 0.0)
        elif kind == "flags":
            R.set_flags(op[1], op[2], op[3], op[4])
        elif kind == "counts":
            R.set_counts(op[1], op[2])
        elif kind == "key":
            R.key(op[1], op[2])  # the reference's handleKeyDown: changes its globals (and traces a frame)
        elif kind == "screen":
            R.set_screen(op[1], op[2])
        elif kind == "background":
            bg = tuple(float(c) for c in op[1])
            R.set_background(bg)
        elif kind == "clear":
            R.clear_scene()
        elif kind == "sphere":
            R.add_sphere(op[1], op[2], op[3], op[4], op[5], *(op[6:8] if len(op) > 6 else (1.0, 0.0)))
        elif kind == "plane":
            R.add_plane(op[1], op[2], op[3], op[4], op[5], *(op[6:8] if len(op) > 6 else (1.0, 0.0)))
        elif kind == "material":
            R.set_material(op[1], op[2], op[3], op[4], op[5])
        elif kind == "light":
            R.add_light(op[1], op[2], list(op[3]))
        elif kind == "camera":
            R.scene_initialize(op[1], op[2], op[3])
        elif kind == "rotate":
            R.rotate_view(op[1], op[2])
        else:
            raise ValueError(kind)
    # frame: pre-filled with the background, then one launchRays()
    frame = np.empty((R.H, R.W, 3))
    frame[:] = np.asarray(bg)
    cap = (R.W + 2) * (R.H + 2)
    log = np.full((cap, 2), -12345, dtype=np.int32)
    t0 = time.time()
    R.lib.ref_launch(frame.ctypes.data_as(rb._dp), log.ctypes.data_as(rb._ip), cap)
    secs = time.time() - t0
    calls, oob = R.lib.ref_sink_calls(), R.out_of_range()
    log = log[:calls]
    idx = R.primary_index(fill=-7)
    st = R.state()
    vals, counts = np.unique(idx, return_counts=True)
    meta = {
        "size": [R.W, R.H], "rgb_fnv": ob.fnv1a64(frame), "idx_fnv": ob.fnv1a64(idx), "calls": int(calls),
        "out_of_range": int(oob), "log_fnv": ob.fnv1a64(log), "min": float(np.nanmin(frame)),
        "max": float(np.nanmax(frame)), "nan": int(np.isnan(frame).sum()),
        "idx_hist": {str(int(v)): int(c) for v, c in zip(vals, counts)},
        "probe": {"%d,%d" % (y, x): [repr(float(c)) for c in frame[y, x]]
                  for (y, x) in ((R.H // 2, R.W // 2), (R.H * 5 // 64, R.W * 25 // 128), (R.H - 1, 0))},
        "state": {k: (v.tolist() if hasattr(v, "tolist") else v) for k, v in st.items()},
        "reference_seconds": round(secs, 3),
    }
    small = R.W * R.H <= 96 * 72
    arrays = {name + "/rgb": frame if small else frame[::16, ::16].copy(),
              name + "/idx": idx if small else idx[::16, ::16].copy()}
    return meta, arrays


def main():
    only = sys.argv[1:]
    out_dir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    metas, arrays = {}, {}
    for name in CASES:
        if only and name not in only:
            continue
        meta, arr = run_case(name)
        metas[name] = meta
        arrays.update(arr)
        print("%-28s %4dx%-4d rgb %s idx %s calls %d oob %d  %.2fs" % (
            name, meta["size"][0], meta["size"][1], meta["rgb_fnv"], meta["idx_fnv"], meta["calls"],
            meta["out_of_range"], meta["reference_seconds"]), flush=True)
    if only:
        return
    with open(os.path.join(out_dir, "golden.json"), "w") as f:
        json.dump({"generator": "oracle/gen_golden.py", "reference": "walker-a/RayTracing compiled in place "
                   "with clang -O3 (the reference Makefile's `accurate` flags)", "cases": metas}, f, indent=1,
                  sort_keys=True)
    np.savez_compressed(os.path.join(out_dir, "frames.npz"), **arrays)


if __name__ == "__main__":
    main()
