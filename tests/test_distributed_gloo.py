"""Frame assembly across ranks (the N > 1 path of bench.py) on CPU: gloo backend, world size 2 and 3.
Band contents are synthetic (rank- and position-coded); no tracing happens here."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from raytracing_amd.distributed import (band_rows, chunk_rows, frame_pieces, gather_bands, local_rows, owned_pieces,
                                        trace_and_gather)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n_rows, n_cols, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        start, count = band_rows(n_rows, world, rank)
        frame = torch.full((n_rows, n_cols, 4), -1.0) if rank == 0 else None
        for step in range(2):  # twice: the buffers are reused frame after frame
            if rank == 0:
                band = frame[start:start + count]
            else:
                band = torch.empty((count, n_cols, 4))
            rows = torch.arange(start, start + count, dtype=torch.float32).view(-1, 1, 1)
            cols = torch.arange(n_cols, dtype=torch.float32).view(1, -1, 1)
            band.copy_((rows * 1000 + cols + step * 0.5).expand(count, n_cols, 4))
            gather_bands(band, frame, n_rows, world, rank)
            dist.barrier()
            if rank == 0:
                want = (torch.arange(n_rows, dtype=torch.float32).view(-1, 1, 1) * 1000 +
                        torch.arange(n_cols, dtype=torch.float32).view(1, -1, 1) + step * 0.5).expand(n_rows, n_cols, 4)
                ret[step] = bool(torch.equal(frame, want))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_rows", [(2, 64), (2, 20), (3, 40), (2, 8)])
def test_gather_bands(world, n_rows):
    port = _free_port()
    with mp.Manager() as m:
        ret = m.dict()
        mp.spawn(_worker, args=(world, port, n_rows, 24, ret), nprocs=world, join=True)
        assert dict(ret) == {0: True, 1: True}


def _pipeline_worker(rank, world, port, n_rows, n_cols, chunks, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        count = sum(n for _, n, _ in owned_pieces(n_rows, world, rank, chunks))
        frame = torch.full((n_rows, n_cols, 4), -1.0) if rank == 0 else None
        band = None if rank == 0 else torch.empty((local_rows(n_rows, world, rank, chunks), n_cols, 4))
        calls = []

        def trace(first, rows, piece):  # stands for the kernel launch: fills the piece's rows
            calls.append((first, rows))
            r = torch.arange(first, first + rows, dtype=torch.float32).view(-1, 1, 1)
            c = torch.arange(n_cols, dtype=torch.float32).view(1, -1, 1)
            piece.copy_((r * 1000 + c).expand(rows, n_cols, 4))

        for _ in range(2):
            trace_and_gather(trace, band, frame, n_rows, world, rank, chunks)
            dist.barrier()
        assert sum(n for _, n in calls) == 2 * count and all(f % 8 == 0 for f, _ in calls)
        if rank == 0:
            want = (torch.arange(n_rows, dtype=torch.float32).view(-1, 1, 1) * 1000 +
                    torch.arange(n_cols, dtype=torch.float32).view(1, -1, 1)).expand(n_rows, n_cols, 4)
            ret[0] = bool(torch.equal(frame, want))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_rows,chunks", [(2, 128, 4), (3, 100, 8), (2, 24, 8), (2, 8, 3), (3, 20, 1)])
def test_trace_and_gather_pipeline(world, n_rows, chunks):
    port = _free_port()
    with mp.Manager() as m:
        ret = m.dict()
        mp.spawn(_pipeline_worker, args=(world, port, n_rows, 16, chunks, ret), nprocs=world, join=True)
        assert dict(ret) == {0: True}


def test_chunk_rows():
    assert chunk_rows(0, 2048, 8) == [(i * 256, 256) for i in range(8)]
    assert chunk_rows(16, 20, 8) == [(16, 8), (24, 8), (32, 4)]
    assert chunk_rows(0, 0, 4) == [] and chunk_rows(8, 5, 4) == [(8, 5)]


def test_frame_pieces():
    for n_rows, world, chunks in [(16384, 8, 8), (100, 3, 8), (8, 2, 3), (20, 3, 1), (513, 4, 2), (1, 2, 2)]:
        pieces = frame_pieces(n_rows, world, chunks)
        assert pieces[0][0] == 0 and sum(n for _, n, _ in pieces) == n_rows
        assert all(a[0] + a[1] == b[0] for a, b in zip(pieces, pieces[1:]))          # contiguous, in row order
        assert all(r0 % 8 == 0 and n > 0 for r0, n, _ in pieces)
        assert [o for _, _, o in pieces] == [k % world for k in range(len(pieces))]  # dealt round-robin
        assert len(pieces) <= world * chunks
        assert sum(local_rows(n_rows, world, r, chunks) for r in range(world)) == n_rows
    assert frame_pieces(16384, 8, 8)[:2] == [(0, 256, 0), (256, 256, 1)]
    assert owned_pieces(64, 2, 1, 2) == [(16, 16, 0), (48, 16, 16)]
