"""Diagnostic: per-wave shader-clock shares of the trace kernel (needs a library built with -DGRAFT_STAMPS).
    GRAFT_LIB=/path/libgraft_stamps.so python profiles/diag_stamps.py cfg3 [height]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracing_amd
from raytracing_amd import scenes
from raytracing_amd.tracer import GRAFT_OUT_F32_RGBA

name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
height = int(sys.argv[2]) if len(sys.argv) > 2 else None
raytracing_amd.init(0)
rt = scenes.workload(name, None, height)
job = rt.create_job()
frame = torch.empty((job.nRows, job.nCols, 4), dtype=torch.float32, device="cuda")
cnt = torch.zeros(24, dtype=torch.int64, device="cuda")
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
for _ in range(2):
    cnt.zero_()
    job.trace(0, job.nRows, GRAFT_OUT_F32_RGBA, frame.data_ptr(), None, cnt.data_ptr(), st.cuda_stream)
    torch.cuda.synchronize()
c = cnt.tolist()
rays, tests, traced = c[:3]
total, hit, rounds, waves, loop = c[4:9]
rays = traced  # lanes busy per round: the queries really made
c = c[:2] + c[4:]  # (the indices below are those of the stamp counters, which start behind GraftCounters)
tiles = ((job.nCols + 7) // 8) * ((job.nRows + 7) // 8)
print("%s: waves %d, tiles %d, wave-rounds %d (%.1f lanes busy per round), cycles per wave %.3g | shares of wave time: "
      "closest-hit %.1f%%, rest of the ray loop %.1f%%, outside the loop (tile set-up, ticket, store) %.1f%% | per round: "
      "closest-hit %.0f cycles, shading/state machine %.0f | per tile outside loop %.0f"
      % (name, waves, tiles, rounds, rays / max(rounds, 1), total / max(waves, 1), 100.0 * hit / total,
         100.0 * (loop - hit) / total, 100.0 * (total - loop) / total, hit / max(rounds, 1), (loop - hit) / max(rounds, 1),
         (total - loop) / tiles))
print("   per wave-round: %.1f third-level entries opened, %.1f groups whose members were screened, %.1f exact sphere tests"
      % (c[9] / max(rounds, 1), c[7] / max(rounds, 1), c[8] / max(rounds, 1)))
if c[12]:
    print("   grid walk: %d calls (%.1f lanes each), %.1f loop iterations per call with %.1f lanes alive on average, "
          "%.0f cycles per call (%.1f%% of wave time), %.1f cycles per iteration"
          % (c[12], c[13] / c[12], c[10] / c[12], c[11] / max(c[10], 1), c[14] / c[12], 100.0 * c[14] / total, c[14] / max(c[10], 1)))
st = c[15:19]  # (stamp counters 17..20 of the kernel, two GraftCounters words taken out above)
if sum(st):
    print("   stages behind the query, shares of wave time: query result %.1f%%, shading %.1f%%, hand-up/sample end %.1f%%, "
          "next ray %.1f%%" % tuple(100.0 * x / total for x in st))
