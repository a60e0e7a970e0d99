"""Where the frame time of a workload goes: the same frame traced with fewer lights, and with the scene moved out of
sight (every tile list empty), on the middle rows of the frame (at most 4096).
    python profiles/decompose_frame.py [cfg4|cfg3|cfg5]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracing_amd
from raytracing_amd import scenes
from raytracing_amd.tracer import GRAFT_OUT_F32_RGBA
raytracing_amd.init(0)
NAME = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
ROWS = 4096


def run(tag, change):
    rt = scenes.workload(NAME)
    change(rt)
    job = rt.create_job()
    rows = min(ROWS, job.nRows)
    buf = torch.empty((rows, job.nCols, 4), dtype=torch.float32, device="cuda")
    cnt = torch.zeros(4, dtype=torch.int64, device="cuda")
    stream = torch.cuda.Stream(); torch.cuda.set_stream(stream); st = stream.cuda_stream
    row0 = (job.nRows - rows) // 2
    job.trace(row0, rows, GRAFT_OUT_F32_RGBA, buf.data_ptr(), None, cnt.data_ptr(), st); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(3):
        job.trace(row0, rows, GRAFT_OUT_F32_RGBA, buf.data_ptr(), None, None, st)
    b.record(); torch.cuda.synchronize()
    c = cnt.tolist()
    print("%-28s level %d  %.3f ms  queries made %.4g (reference %.4g)" % (tag, job.screen_level(), a.elapsed_time(b) / 3, c[2], c[0]), flush=True)
    job.close()


def lights(n):
    def f(rt):
        rt.numLights = n
    return f


def away(rt):
    for s in rt.shapes[:rt.numShapes]:
        s.unif[1] += 5000.0   # far above the view: no tile sees a sphere
run("4 lights (the workload)", lights(4))
run("3 lights", lights(3))
run("2 lights", lights(2))
run("1 light", lights(1))
run("no light", lights(0))
run("scene out of sight", away)
