"""cfg5 traced with maxDepth 1 (no mirror rays are shot) and 2: what the rays behind the mirrors cost.
    python profiles/cfg5_split.py"""
import sys, torch
sys.path.insert(0, "/root/repo")
import raytracing_amd
from raytracing_amd import scenes
from raytracing_amd.tracer import GRAFT_OUT_F32_RGBA
raytracing_amd.init(0)
for depth in (1, 2):
    rt = scenes.workload("cfg5")
    rt.maxDepth = depth
    job = rt.create_job()
    buf = torch.empty((job.nRows, job.nCols, 4), dtype=torch.float32, device="cuda")
    cnt = torch.zeros(4, dtype=torch.int64, device="cuda")
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    st = stream.cuda_stream
    job.trace(0, job.nRows, GRAFT_OUT_F32_RGBA, buf.data_ptr(), None, cnt.data_ptr(), st)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        job.trace(0, job.nRows, GRAFT_OUT_F32_RGBA, buf.data_ptr(), None, None, st)
    b.record(); torch.cuda.synchronize()
    print("cfg5 maxDepth", depth, "level", job.screen_level(), "%.3f ms" % (a.elapsed_time(b) / 5), "queries", cnt.tolist())
    job.close()
