#!/usr/bin/env python3
"""bench.py -- throughput of the launchRays() path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg4|cfg3|cfg5|default|single]

One step = one whole frame of the workload: every rank traces its rows with the HIP kernel
(inputs -- packed scene, loop-offset tables -- already resident in HBM) into a float4 framebuffer; for
N > 1 the rows are dealt to the ranks in pieces and every piece is sent to rank 0's HBM over RCCL while the next one
is being traced.  Prints ONE JSON line on rank 0.

metric  Mrays/s, ray = one closest-hit query (primary + shadow + reflection; rayIntersectAvoid of the
        reference, mainTracing.c:128) that the library MADE, counted by the kernel itself in an untimed pass.  The
        reference makes more of them for the same frame (shadow rays whose answer its lighting() overwrites); that
        count and the rate it would give are under `reference`, the frame time is ms_per_step.
roofline  the trace kernel is bound by the vector ALU.  achieved = FP64 flop the kernel EXECUTED per launch (hardware
        counters of the same command, kept under profiles/ because counters cannot be read in-process:
        (ADD + MUL + 2 FMA + TRANS wave-instructions) x 64 lanes) / mean kernel time measured here (HIP events on the
        launch stream, inside the timed region); peak 78.6 TFLOP/s = 256 CU x 4 SIMD x 16 FP64 lanes x 2 flop (FMA)
        x 2.4 GHz.  Most of the bit-exact arithmetic cannot fuse multiply and add (only divisions, square roots and the
        exact loop's proof of misses use FMAs), so `fp64_issue_frac` is given beside it: the share of the vector ALU's
        issue slots (39.3e12 instruction-lanes per second) taken by FP64 arithmetic instructions, at most 1 whatever
        the mix.  `algorithmic` holds the other view: 17 flop of the reference's brute-force test for every
        (ray, shape) pair / kernel time -- the screens prove most pairs to be misses without executing them, so that
        rate can exceed the peak; it is a speed-up over brute force, not a utilisation.
cpu_baseline  the CPU oracle (a port of the reference, bit-exact to it) on a bounded sample of rows of the
        SAME frame, on `cores` host threads; the sampled rows are also compared with the GPU's.
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import raytracing_amd  # noqa: E402
from raytracing_amd import scenes  # noqa: E402
from raytracing_amd.distributed import local_rows, trace_and_gather  # noqa: E402
from raytracing_amd.tracer import GRAFT_OUT_F32_RGBA, GRAFT_OUT_F64_RGB, GRAFT_OUT_U8_RGBA  # noqa: E402

FLOP_PER_TEST = 17            # FP64 flops of one ray-sphere test on the miss path (shapes.c:52-54)
PEAK_FP64_VALU = 78.6         # TFLOP/s, FMA counted as 2
PEAK_FP32_VALU = 157.3        # TFLOP/s, packed FMA


def build_workload(name, width=None, height=None):
    if name in scenes.WORKLOADS:
        rt = scenes.workload(name, width, height)
        w = scenes.WORKLOADS[name]
        desc = "%s: %dx%d, %d random spheres, 4 lights, %s, maxDepth %d" % (
            name, rt.WIDTH, rt.HEIGHT, w["n"],
            ("%dx%d regular-grid supersampling" % (w["passes"], w["passes"])) if w["passes"] else "1 spp", w["depth"])
    elif name == "default":
        rt = scenes.default_scene(WIDTH=width or 512, HEIGHT=height or 512)
        desc = "default: reference start-up scene (11 shapes, 1 light), 6x6 supersampling, maxDepth 4"
    elif name == "single":
        rt = scenes.single_sphere_scene(aliasing=0, WIDTH=width or 512, HEIGHT=height or 512)
        desc = "single: 512x512, shape 0 + light 0 of the start-up scene, 1 spp"
    else:
        raise SystemExit("unknown workload %r" % name)
    return rt, desc


def cpu_baseline(rt, gpu_band, row0, n_rows, cores, budget_rows):
    """Oracle on `budget_rows` rows spread over [row0,row0+n_rows); returns the cpu_baseline object."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_binding as ob
    picks = sorted(set(int(row0 + (i + 0.5) * n_rows / budget_rows) for i in range(budget_rows)))
    rays = 0
    secs = 0.0
    exact = True
    for r in picks:
        t0 = time.perf_counter()
        o = ob.render(rt, r, 1, threads=cores, want_counters=True)
        secs += time.perf_counter() - t0
        rays += o["counters"][0]
        if gpu_band is not None:
            mine = gpu_band[r - row0, :, :3].cpu().numpy()
            if mine.dtype == np.uint8:  # RGBA8 frame: the quantiser of the RGB bar, q(v) = (uint8)(min(max(v,0),1)*255 + 0.5)
                want = (np.clip(np.nan_to_num(o["rgb"][0]), 0, 1) * 255 + 0.5).astype(np.uint8)
            else:
                want = o["rgb"][0].astype(np.float32)
            exact = exact and np.array_equal(mine, want)
    # the reference's own execution model is one thread: the same port on one core, over a few of the sampled rows
    few = picks[::max(1, len(picks) // 4)][:4]
    rays1, secs1 = 0, 0.0
    for r in few:
        t0 = time.perf_counter()
        o = ob.render(rt, r, 1, threads=1, want_counters=True)
        secs1 += time.perf_counter() - t0
        rays1 += o["counters"][0]
    return {"value": rays / secs / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": "%d of %d rows of the same frame (evenly spaced), %.1f s of CPU work" % (len(picks), rt.HEIGHT, secs),
            "whole_frame_s_extrapolated": secs * n_rows / len(picks),
            "sample_matches_gpu": bool(exact),
            "single_core_value": rays1 / secs1 / 1e6, "single_core_sample": "%d of those rows, %.1f s" % (len(few), secs1)}


def cpu_reference(name):
    """The REAL reference (oracle/_ref: its own sources compiled for a 512x512 frame) tracing the workload's scene,
    one sample per pixel, on one host core -- the reference is single-threaded.  The frame is compared with the GPU's
    of the same 512x512 job, whose ray count (the kernel's own counter) gives the rate.  None where the reference
    build is absent."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ref_binding as rb
    if name not in scenes.WORKLOADS or not rb.available("librefp_512x512.so"):
        return None
    rt = scenes.workload(name, 512, 512)
    rt.aliasing, rt.numPasses = 0, 1
    R = rb.Reference((512, 512))
    if rt.numShapes > R.max_shapes or rt.numLights > R.max_lights:
        return None
    R.clear_scene()
    R.set_screen(rt.screenWidth, rt.screenHeight)
    R.set_background(rt.backgroundColor)
    for s in rt.shapes[:rt.numShapes]:
        R.add_sphere(s.unif[0:3], s.unif[3], s.reflectivity, s.unif[4:7], s.ambientLight, s.refrIndex, s.transparency)
    for l in rt.lights[:rt.numLights]:
        R.add_light(l.color, l.loc, [])
    R.scene_initialize(rt.target, rt.targetToScreen, rt.screenToCam)
    R.set_flags(rt.projectionType, 0, 1, rt.maxDepth)
    t0 = time.perf_counter()
    frame = R.launch(fill=0.0)
    secs = time.perf_counter() - t0
    rt.render_rows(0, None, GRAFT_OUT_F64_RGB, want_counters=True)
    rays = rt.last_counters[0]
    same = bool(np.array_equal(rt.launchRays(), frame))  # launchRays replays pixSetRGB as the reference shows it
    return {"value": rays / secs / 1e6, "unit": "Mrays/s", "cores": 1, "kind": "reference",
            "sample": "the workload's scene at 512x512, 1 sample per pixel, through the reference's own launchRays "
                      "(%.2f s)" % secs, "frame_matches_gpu": same}


def frame_512(dev):
    """The metric's "512x512 frame time" and what the host pays around the kernel.  Per scene (the reference's
    start-up scene cut to one sphere + one light at 1 spp; the full start-up scene with 6x6 supersampling; the cfg3
    scene at 1 spp): kernel time (HIP events), graftJobCreate (pack + upload) on a first call and on a second call
    that finds the screen tables in the library's cache, and one whole launchRays-style call with host buffers
    (graftRenderRows into a page-locked frame: pack + upload + trace + copy back over PCIe) on the first call and
    when scene and view are unchanged since the call before."""
    import ctypes
    from raytracing_amd._lib import check, lib
    out = {}
    stream = torch.cuda.current_stream()

    def cfg3_512():
        rt = scenes.workload("cfg3", 512, 512)
        return rt

    for key, make in (("single_sphere", lambda: build_workload("single")[0]),
                      ("startup_scene_aa6", lambda: build_workload("default")[0]), ("cfg3_scene_512", cfg3_512)):
        rt = make()
        t0 = time.perf_counter()
        job = rt.create_job()
        out[key + "_job_create_first_ms"] = (time.perf_counter() - t0) * 1e3
        job.close()
        t0 = time.perf_counter()
        job = rt.create_job()
        out[key + "_job_create_again_ms"] = (time.perf_counter() - t0) * 1e3
        buf = torch.empty((job.nRows, job.nCols, 4), dtype=torch.float32, device=dev)
        for _ in range(3):
            job.trace(0, job.nRows, GRAFT_OUT_F32_RGBA, buf.data_ptr(), None, None, stream.cuda_stream)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        a.record()
        for _ in range(reps):
            job.trace(0, job.nRows, GRAFT_OUT_F32_RGBA, buf.data_ptr(), None, None, stream.cuda_stream)
        b.record()
        torch.cuda.synchronize()
        out[key + "_kernel_ms"] = a.elapsed_time(b) / reps
        job.close()
        # whole host calls: the doubles pixSetRGB receives (24 bytes per pixel), into page-locked memory
        scene, view, keep = rt.pack()
        nbytes = view.nRows * view.nCols * 24
        host = ctypes.c_void_p()
        check(lib.graftHostAlloc(nbytes, ctypes.byref(host)))
        call = lambda: check(lib.graftRenderRows(ctypes.byref(scene), ctypes.byref(view), 0, view.nRows,
                                                 GRAFT_OUT_F64_RGB, host, None, None))
        other = build_workload("single" if key != "single_sphere" else "default")[0]
        other.render_rows(0, 8)  # puts another job in the library's one-entry cache
        t0 = time.perf_counter()
        call()
        out[key + "_host_call_first_ms"] = (time.perf_counter() - t0) * 1e3
        t0 = time.perf_counter()
        for _ in range(5):
            call()
        out[key + "_host_call_unchanged_ms"] = (time.perf_counter() - t0) / 5 * 1e3
        job = rt.create_job()
        f64 = torch.empty((view.nRows, view.nCols, 3), dtype=torch.float64, device=dev)
        job.trace(0, view.nRows, GRAFT_OUT_F64_RGB, f64.data_ptr(), None, None, stream.cuda_stream)
        a.record()
        for _ in range(reps):
            job.trace(0, view.nRows, GRAFT_OUT_F64_RGB, f64.data_ptr(), None, None, stream.cuda_stream)
        b.record()
        torch.cuda.synchronize()
        out[key + "_kernel_f64_ms"] = a.elapsed_time(b) / reps
        job.close()
        lib.graftHostFree(host)
        del keep
    return out


def host_frame(rt, kernel_ms):
    """What a host caller pays for the whole frame of the measured workload: graftRenderRows into page-locked host
    memory (scene and view unchanged since the call before: nothing is packed), per output format.  A large frame is
    traced in pieces of rows whose copies to the host run beside the tracing of the next pieces; `one_piece` times
    the same call with that switched off (kernel, then one copy)."""
    import ctypes
    from raytracing_amd._lib import check, lib
    from raytracing_amd.tracer import GRAFT_OUT_U8_RGBA
    scene, view, keep = rt.pack()
    out = {"kernel_ms": kernel_ms}
    px = view.nRows * view.nCols
    for key, fmt, size in (("u8_rgba", GRAFT_OUT_U8_RGBA, 4), ("f32_rgba", GRAFT_OUT_F32_RGBA, 16), ("f64_rgb", GRAFT_OUT_F64_RGB, 24)):
        host = ctypes.c_void_p()
        check(lib.graftHostAlloc(px * size, ctypes.byref(host)))
        try:
            call = lambda: check(lib.graftRenderRows(ctypes.byref(scene), ctypes.byref(view), 0, view.nRows, fmt, host, None, None))
            call()  # (first call: packs, uploads, sizes the device-side frame)
            t0 = time.perf_counter()
            for _ in range(2):
                call()
            out[key + "_ms"] = (time.perf_counter() - t0) / 2 * 1e3
            old = raytracing_amd.set_diagnostic("copy_piece_kb", 0)
            call()
            t0 = time.perf_counter()
            call()
            out[key + "_one_piece_ms"] = (time.perf_counter() - t0) * 1e3
            raytracing_amd.set_diagnostic("copy_piece_kb", old)
            out[key + "_bytes"] = px * size
        finally:
            lib.graftHostFree(host)
    del keep
    return out


KERNEL_SOURCES = ("grafthip.hip", "closest_hit.h", "trace_types.h", "pack_tables.h", "host_job.h")


def kernel_source_hash():
    """Fingerprint of the sources libgrafthip.so is built from: a counter record belongs to one build."""
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "raytracing_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def recorded_counters(workload, frame, screen_cap):
    """What the hardware counters said about this same command (profiles/collect_pmc.sh + summarize_pmc.py):
    profiles/pmc_<workload>[_level<cap>].json.  Returns (record, note): the record is used only when it was taken
    on this frame size; the note says whether it belongs to the build that is running."""
    name = "pmc_%s%s.json" % (workload, "" if screen_cap >= 7 else "_level%d" % screen_cap)
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            rec = json.load(f)
    except (OSError, ValueError):
        return None, "no counter record profiles/%s" % name
    if rec.get("frame") != frame:
        return None, "profiles/%s was taken on frame %s" % (name, rec.get("frame"))
    same = rec.get("kernel_source_hash") == kernel_source_hash()
    return rec, "profiles/%s, %s" % (name, "same kernel sources" if same else
                                     "RECORDED ON AN OLDER BUILD (%s)" % rec.get("kernel_source_hash"))


def roofline(rec, note, kernel_ms, local_tests):
    """Executed-work roofline of the trace kernel from the recorded counters and the kernel time measured now."""
    secs = kernel_ms * 1e-3
    algorithmic = FLOP_PER_TEST * local_tests / secs / 1e12
    out = {"bound": "valu_fp64", "achieved": None, "peak": PEAK_FP64_VALU, "unit": "TFLOP/s", "frac": None,
           "fp64_issue_frac": None, "traffic": None, "kernel": "traceKernel", "kernel_ms": kernel_ms,
           "counters": note,
           "algorithmic": {"tflops": algorithmic, "flop_per_test": FLOP_PER_TEST, "tests_per_launch": local_tests,
                           "speedup_over_fp64_brute_force_ceiling": algorithmic / PEAK_FP64_VALU,
                           "note": "17 flop of the reference's test for every (ray, shape) pair; pairs the screens "
                                   "prove to be misses are counted although they are never executed"}}
    if rec:
        c = rec["counters"]
        f64 = (c["SQ_INSTS_VALU_ADD_F64"] + c["SQ_INSTS_VALU_MUL_F64"] + 2 * c["SQ_INSTS_VALU_FMA_F64"]
               + c["SQ_INSTS_VALU_TRANS_F64"]) * 64
        f32 = (c["SQ_INSTS_VALU_ADD_F32"] + c["SQ_INSTS_VALU_MUL_F32"] + 2 * c["SQ_INSTS_VALU_FMA_F32"]) * 64
        fp_instr = sum(c[k] for k in c if k.startswith("SQ_INSTS_VALU_") and k.endswith(("_F64", "_F32")))
        out.update({
            "achieved": f64 / secs / 1e12, "frac": f64 / secs / 1e12 / PEAK_FP64_VALU,
            "fp64_issue_frac": (c["SQ_INSTS_VALU_ADD_F64"] + c["SQ_INSTS_VALU_MUL_F64"] + c["SQ_INSTS_VALU_FMA_F64"]
                                + c["SQ_INSTS_VALU_TRANS_F64"]) * 64 / secs / (PEAK_FP64_VALU / 2 * 1e12),
            "executed_fp64_flop_per_launch": f64,
            "executed_fp32_tflops": {"unpacked": f32 / secs / 1e12, "if_all_packed": 2 * f32 / secs / 1e12,
                                     "peak": PEAK_FP32_VALU},
            "fp_share_of_valu_instructions": fp_instr / c["SQ_INSTS_VALU"],
            "valu_busy": rec["derived"].get("valu_busy_fraction_per_simd"),
            "lane_utilisation": rec["derived"].get("lane_utilisation"),
            "valu_instructions_per_wave_ray": rec["derived"].get("valu_wave_instr_per_wave_ray"),
            "lds_busy": rec["derived"].get("lds_busy_fraction"),
            "clock_ghz_under_profile": rec["derived"].get("clock_ghz"),
            "traffic": rec["derived"].get("hbm_bytes_per_launch"),
            "recorded_kernel_ms": rec["derived"].get("kernel_ms"),
        })
    return out


def host_threads():
    """Host threads this process may use: the box's share (OMP_NUM_THREADS is set to it there), else the CPUs this
    process is allowed on."""
    try:
        n = int(os.environ.get("OMP_NUM_THREADS", "0"))
    except ValueError:
        n = 0
    return n if n > 0 else len(os.sched_getaffinity(0))


def rank_diagnostics(per_step, steps, local_rays, step_ms, world, dev):
    """What every rank did per frame: its rays, the device time of its own kernels (first launch to last kernel done,
    the transfers queued between them included), the device time from its last kernel's end to its last transfer's
    end (the part of the exchange that no tracing on that rank hid), the host time spent after the last kernel was
    queued, and the whole step.  Collective: every rank calls it."""
    mine = torch.tensor([local_rays,
                         sum(d["trace_event"][0].elapsed_time(d["trace_event"][1]) for d in per_step) / steps,
                         sum(d["wait_s"] for d in per_step) / steps * 1e3, step_ms,
                         sum(d["pieces"] for d in per_step) / steps,
                         sum(d["tail_event"][0].elapsed_time(d["tail_event"][1]) for d in per_step) / steps],
                        dtype=torch.float64, device=dev)
    gathered = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    return [{"rank": r, "rays": int(g[0]), "trace_ms": float(g[1]), "transfer_wait_ms": float(g[2]),
             "step_ms": float(g[3]), "pieces": int(g[4]), "last_kernel_to_last_transfer_ms": float(g[5])}
            for r, g in enumerate(gathered)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cfg4")
    ap.add_argument("--width", type=int, default=None, help="reduced frame width (debugging)")
    ap.add_argument("--height", type=int, default=None, help="reduced frame height (debugging)")
    ap.add_argument("--screen", type=int, default=7, help="cap of the screen level (0 = the exact double loop only)")
    ap.add_argument("--cpu-rows", type=int, default=None, help="rows of the frame the CPU baseline traces")
    ap.add_argument("--cpu-cores", type=int, default=None, help="host threads of the CPU baseline (default: all this "
                    "process may use)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-frame512", action="store_true")
    ap.add_argument("--no-host-frame", action="store_true")
    ap.add_argument("--chunks", type=int, default=None,
                    help="pieces per GPU in the trace/gather pipeline (N > 1); default 8 * sqrt(8 / N)")
    ap.add_argument("--format", choices=("f32", "u8"), default="f32",
                    help="frame in HBM: float4 RGBA (16 bytes per pixel, the default) or RGBA8 (4 bytes: a quarter of the "
                         "bytes the gather moves at N > 1)")
    ap.add_argument("--reserve-cus", type=int, default=None,
                    help="compute units every trace leaves free (default: 4 at N > 1, where RCCL's transfer kernels "
                         "run beside the traces; 0 on one GPU)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.chunks is None:
        # Every piece is a launch of its own and ends with a tail of idle lanes (about half a tile's time, 0.14 ms
        # on the default workload); the transfer of a rank's last piece is not hidden behind any tracing.  With B
        # bytes per rank the sum 0.14 ms * p + B / p / link rate is smallest near p = 8 at N = 8 and grows with
        # the square root of the bytes per rank.
        args.chunks = max(4, int(round(8.0 * (8.0 / max(world, 1)) ** 0.5)))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d"
                         % (args.gpus, world, args.gpus))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    raytracing_amd.init(local)
    raytracing_amd.set_diagnostic("screen_cap", args.screen)
    if args.reserve_cus is None:
        args.reserve_cus = 4 if world > 1 else 0
    raytracing_amd.set_diagnostic("reserved_cus", args.reserve_cus)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

    rt, desc = build_workload(args.workload, args.width, args.height)
    job = rt.create_job()
    W, H = job.nCols, job.nRows
    row0, n_rows = 0, H
    fmt, frame_dtype = (GRAFT_OUT_U8_RGBA, torch.uint8) if args.format == "u8" else (GRAFT_OUT_F32_RGBA, torch.float32)
    if rank == 0:  # rank 0 owns the assembled frame and traces straight into its own rows of it
        frame = band = torch.empty((H, W, 4), dtype=frame_dtype, device=dev)
    else:          # the other ranks trace their pieces into a buffer of their own and send them to rank 0
        frame = None
        band = torch.empty((local_rows(H, world, rank, args.chunks), W, 4), dtype=frame_dtype, device=dev)
    counters = torch.zeros(4, dtype=torch.int64, device=dev)  # one GraftCounters
    # a real (non-null) stream: kernels, events and the RCCL transfers of the gather are all ordered on it
    torch_stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(torch_stream)
    stream = torch_stream.cuda_stream

    def step(count=False, stats=None):
        def trace(first, rows, piece):
            job.trace(first, rows, fmt, piece.data_ptr(), None, counters.data_ptr() if count else None, stream)
        trace_and_gather(trace, band, frame, H, world, rank, args.chunks, stats=stats)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # untimed: ray statistics of one frame (the kernel counts its own closest-hit queries and tests)
    step(count=True)
    torch.cuda.synchronize()
    stats = counters.clone()
    if world > 1:
        dist.all_reduce(stats)
    # stats: GraftCounters = the reference's queries and tests for this frame, and the queries the library made
    ref_rays, tests, rays = int(stats[0]), int(stats[1]), int(stats[2])
    local_rays, local_tests = int(counters[2]), int(counters[1])

    for _ in range(args.warmup):
        step()
    fence()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    per_step = [dict() for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        if world == 1:  # one launch per frame: its HIP-event time is the kernel time of the roofline
            ev[k][0].record()
            job.trace(row0, n_rows, fmt, band.data_ptr(), None, None, stream)
            ev[k][1].record()
        else:           # pieces of the band are traced and shipped to rank 0 in a pipeline
            ev[k][0].record()
            step(stats=per_step[k])
            ev[k][1].record()
    fence()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t[0])
    kernel_ms = sum(a.elapsed_time(b) for a, b in ev) / args.steps

    per_rank, per_rank_error = None, None
    if world > 1:
        try:  # diagnostics only: whatever goes wrong here must not cost the run its result line
            per_rank = rank_diagnostics(per_step, args.steps, local_rays, kernel_ms, world, dev)
        except Exception as e:  # noqa: BLE001
            per_rank_error = repr(e)

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        rec, note = recorded_counters(args.workload, [W, H], args.screen)
        out = {
            "metric": "Mrays/s", "value": rays / (elapsed / args.steps) / 1e6, "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic", "config": {"workload": desc, "frame": [W, H],
                                            "output": "RGBA8 in HBM" if args.format == "u8" else "float4 RGBA in HBM",
                                            "screen_level": job.screen_level(),
                                            "rays_counted": "closest-hit queries the kernel made (%.4g per frame); the "
                                                            "reference makes %.4g for the same frame, see `reference`"
                                                            % (rays, ref_rays),
                                            "parallelism": ("1 GPU" if world == 1 else "rows dealt round-robin in %d pieces per GPU x%d, "
                                                            "RCCL send/recv to rank 0 queued behind each piece, "
                                                            "%d CUs per GPU left free for them"
                                                            % (args.chunks, world, args.reserve_cus)),
                                            "reserved_cus": args.reserve_cus},
            "rays_per_frame": rays, "tests_per_frame": tests,
            "reference": {"rays_per_frame": ref_rays,
                          "equivalent_Mrays_per_s": ref_rays / (elapsed / args.steps) / 1e6,
                          "note": "closest-hit queries the reference makes for this frame.  `value` counts the "
                                  "queries the library made: per shaded hit it asks the lights last to first and "
                                  "stops at the first unshadowed one, because the reference's lighting() lets every "
                                  "unshadowed light overwrite the one before (mainTracing.c:223-225); frames are "
                                  "bit-identical, and tests_per_frame (the roofline's algorithmic count) is the "
                                  "reference's"},
            "primary_Msamples_per_s": W * H * (rt.numPasses ** 2 if rt.aliasing else 1) / (elapsed / args.steps) / 1e6,
            "kernel_source_hash": kernel_source_hash(),
        }
        if world == 1:
            out["roofline"] = roofline(rec, note, kernel_ms, local_tests)
        else:  # the per-launch kernel time is not isolated here; the 1-GPU line carries the roofline
            out["roofline"] = {"bound": "valu_fp64", "achieved": None, "peak": PEAK_FP64_VALU, "unit": "TFLOP/s",
                               "frac": None, "traffic": None, "kernel": "traceKernel",
                               "note": "see the 1-GPU line: at N > 1 a step is several launches per rank"}
            out["per_rank"] = per_rank if per_rank is not None else {"error": per_rank_error}
            if per_rank:
                out["ray_imbalance"] = max(p["rays"] for p in per_rank) / (rays / world)
            torch.cuda.synchronize()
            # the assembled frame against rank 0's own trace of 64 tile rows spread over all pieces
            check = torch.empty((8, W, 4), dtype=frame_dtype, device=dev)
            same, checked = True, 0
            for i in range(64):
                r = min(int((i + 0.5) * H / 64) // 8 * 8, max(H - 8, 0))
                n = min(8, H - r)
                job.trace(r, n, fmt, check.data_ptr(), None, None, stream)
                torch.cuda.synchronize()
                same = same and bool(torch.equal(check[:n], frame[r:r + n]))
                checked += 1
            out["gathered_frame_matches_local_trace"] = same
            out["gathered_frame_tile_rows_checked"] = checked
        if world == 1 and not args.no_frame512:
            out["frame_512"] = frame_512(dev)
        if world == 1 and not args.no_host_frame:
            out["host_frame"] = host_frame(rt, kernel_ms)
        if world == 1 and not args.no_cpu:
            cores = args.cpu_cores or host_threads()
            rows = args.cpu_rows
            if rows is None:  # aim at roughly 15 s of CPU work: ~4e8 ray-shape tests/s per host core
                per_row = max(tests / H, 1.0)
                rows = int(max(1, min(H, 15.0 * 4.0e8 * cores / per_row)))
            out["cpu_baseline"] = cpu_baseline(rt, band, row0, n_rows, cores, rows)
            out["cpu_baseline"]["host_cpus_visible"] = os.cpu_count()
            ref = cpu_reference(args.workload)
            if ref is not None:
                out["cpu_reference"] = ref
        print(json.dumps(out), flush=True)
    job.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
