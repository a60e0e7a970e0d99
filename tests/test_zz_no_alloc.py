"""graftRenderRowsNoAlloc (the entry point that allocates nothing: its device block is a static array of the library)
against graftRenderRows on the same inputs, bit for bit, through the C ABI: two calls in a row, the second with a
smaller input, then a call with other rows and another format; an input beyond the static block is refused."""
import numpy as np
import pytest
import torch

from raytracing_amd import GRAFT_OUT_F32_RGBA, GRAFT_OUT_F64_RGB, GraftError, RayTracer, scenes
from raytracing_amd._lib import check, lib

pytestmark = pytest.mark.gpu


def both(rt, row0, n_rows, fmt=GRAFT_OUT_F64_RGB):
    want_rgb, want_idx = rt.render_rows(row0, n_rows, fmt, want_index=True, want_counters=True)
    want_counters = rt.last_counters
    dev = torch.device("cuda", 0)
    dtype, ch = (torch.float64, 3) if fmt == GRAFT_OUT_F64_RGB else (torch.float32, 4)
    frame = torch.full((n_rows, want_rgb.shape[1], ch), -1.0, dtype=dtype, device=dev)
    index = torch.full((n_rows, want_rgb.shape[1]), -9, dtype=torch.int32, device=dev)
    counters = torch.zeros(4, dtype=torch.int64, device=dev)  # one GraftCounters
    torch.cuda.synchronize()
    rt.render_rows_no_alloc(row0, n_rows, fmt, frame.data_ptr(), index.data_ptr(), counters.data_ptr())
    check(lib.graftSynchronize())
    got = frame.cpu().numpy()
    assert np.array_equal(got.view(np.uint64 if fmt == GRAFT_OUT_F64_RGB else np.uint32),
                          want_rgb.view(np.uint64 if fmt == GRAFT_OUT_F64_RGB else np.uint32))
    assert np.array_equal(index.cpu().numpy(), want_idx)
    assert tuple(int(c) for c in counters.cpu()[:2]) == tuple(want_counters)
    assert int(counters[2]) == rt.last_rays_traced


def test_no_alloc_entry_matches_render_rows(gpu):
    import test_gpu_screens as ts
    big = ts.scene(50, 300, size=(96, 96), lights=3, refl=0.5, depth=4)       # per-family screens in LDS
    both(big, 0, 96)
    small = scenes.default_scene(aliasing=1, numPasses=2, maxDepth=4, WIDTH=64, HEIGHT=40)  # planes, fewer tables
    both(small, 0, 40)
    both(big, 17, 30, GRAFT_OUT_F32_RGBA)                                     # other rows, other format
    both(scenes.workload("cfg5", 64, 64), 8, 48)                              # 4096 spheres: tables in global memory


def test_no_alloc_entry_refuses_what_does_not_fit(gpu):
    rt = RayTracer(700000, 8)  # 5.6 MB of loop offsets alone
    rt.sceneInitialize([0, 0, 0], 256, 1000)
    frame = torch.zeros(8, dtype=torch.float64, device="cuda:0")
    with pytest.raises(GraftError) as e:
        rt.render_rows_no_alloc(0, 8, GRAFT_OUT_F64_RGB, frame.data_ptr())
    assert e.value.code == 2 and "scratch" in str(e.value)
    both(scenes.single_sphere_scene(aliasing=0, WIDTH=64, HEIGHT=64), 0, 64)  # the entry point still works afterwards
