"""Row-band sharding of one frame over the GPUs of a node (one process per GPU, torch.distributed).

Pixels are independent and the scene is tiny, so every rank holds the whole packed scene and traces a
contiguous band of loop rows; the frame is row-major in y, so a band is one contiguous block.  The only
exchange is the assembly of the frame in rank 0's memory: every other rank sends its band straight to
rank 0 (a grouped send/recv, which is how RCCL implements a gather; on the xGMI mesh each sender has its
own link to the root, so the transfers run concurrently).  Rank 0 traces directly into its rows of the
frame and sends nothing.
"""
import torch.distributed as dist

TILE_ROWS = 8  # bands are cut at multiples of the kernel's 8x8 pixel tile


def band_rows(n_rows, world, rank):
    """(first row, row count) of `rank`'s band: equal tile-aligned bands, the last one takes what is left
    (possibly nothing when there are more ranks than tile rows)."""
    tiles = (n_rows + TILE_ROWS - 1) // TILE_ROWS
    per = (tiles + world - 1) // world * TILE_ROWS
    start = min(rank * per, n_rows)
    stop = min(start + per, n_rows)
    return start, stop - start


def gather_bands(band, frame, n_rows, world, rank, group=None):
    """Assemble the frame on rank 0.  `band` is this rank's [rows][cols][channels] tensor; `frame` is the
    whole [n_rows][cols][channels] tensor on rank 0 (ignored elsewhere).  On rank 0 `band` should be the view
    frame[start:start+count] so that its own rows need no copy; if it is not, it is copied in."""
    if world == 1:
        if frame is not None and band.data_ptr() != frame.data_ptr():
            frame[: band.shape[0]].copy_(band)
        return
    ops = []
    if rank == 0:
        start, count = band_rows(n_rows, world, 0)
        own = frame[start:start + count]
        if count and own.data_ptr() != band.data_ptr():
            own.copy_(band)
        for r in range(1, world):
            start, count = band_rows(n_rows, world, r)
            if count:
                ops.append(dist.P2POp(dist.irecv, frame[start:start + count], r, group))
    elif band.shape[0]:
        ops.append(dist.P2POp(dist.isend, band, 0, group))
    if ops:
        for work in dist.batch_isend_irecv(ops):
            work.wait()


def chunk_rows(start, count, chunks):
    """Cut rows [start, start+count) into at most `chunks` tile-aligned pieces: list of (first row, rows)."""
    if count <= 0:
        return []
    tiles = (count + TILE_ROWS - 1) // TILE_ROWS
    chunks = max(1, min(chunks, tiles))
    per = (tiles + chunks - 1) // chunks * TILE_ROWS
    out = []
    r = start
    while r < start + count:
        n = min(per, start + count - r)
        out.append((r, n))
        r += n
    return out


def trace_and_gather(trace, band, frame, n_rows, world, rank, chunks=8, group=None):
    """One frame: trace this rank's band piece by piece and ship every finished piece to rank 0 while the next one
    is being traced.  `trace(first_row, rows, tensor)` must queue the kernel that fills `tensor` (the piece's
    [rows][cols][channels] view) on the current stream.  The transfers are queued behind the piece's kernel (RCCL
    waits for the current stream at the time of the call) and only waited for at the end of the frame, so on a
    strong-scaled frame the exchange hides behind the tracing of the later pieces.  Every rank cuts every band the
    same way, so sends and receives pair up in order."""
    start, count = band_rows(n_rows, world, rank)
    works = []
    if world == 1:
        for r0, n in chunk_rows(start, count, 1):
            trace(r0, n, band[r0 - start:r0 - start + n])
        if frame is not None and band.data_ptr() != frame.data_ptr():
            frame[:count].copy_(band)
        return
    mine = chunk_rows(start, count, chunks)
    others = {r: chunk_rows(*band_rows(n_rows, world, r), chunks) for r in range(1, world)} if rank == 0 else {}
    rounds = max([len(mine)] + [len(v) for v in others.values()])
    for c in range(rounds):
        ops = []
        if c < len(mine):
            r0, n = mine[c]
            piece = band[r0 - start:r0 - start + n]
            trace(r0, n, piece)
            if rank != 0:
                ops.append(dist.P2POp(dist.isend, piece, 0, group))
        if rank == 0:
            for r, pieces in others.items():
                if c < len(pieces):
                    r0, n = pieces[c]
                    ops.append(dist.P2POp(dist.irecv, frame[r0:r0 + n], r, group))
        if ops:
            works.extend(dist.batch_isend_irecv(ops))
    for w in works:
        w.wait()
