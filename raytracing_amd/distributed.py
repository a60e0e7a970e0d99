"""Row sharding of one frame over the GPUs of a node (one process per GPU, torch.distributed).

Pixels are independent and the scene is tiny, so every rank holds the whole packed scene and traces rows of the
frame; the frame is row-major in y, so a run of rows is one contiguous block.  The only exchange is the assembly of
the frame in rank 0's memory: every other rank sends its rows straight to rank 0 (grouped send/recv, which is how
RCCL implements a gather; on the xGMI mesh each sender has its own link to the root, so the transfers run
concurrently).  Rank 0 traces directly into its rows of the frame and sends nothing.

Two decompositions: `band_rows` (one contiguous band per rank, `gather_bands`) and `frame_pieces` (the frame cut
into world x chunks pieces dealt round-robin, `trace_and_gather`).  The second is what bench.py uses: rows differ in
cost (how many spheres their rays see), and dealing pieces of every part of the frame to every rank evens that out
(ray counts of the 1024-sphere workload on 8 ranks: largest contiguous band 11 % above the mean, round-robin 3 %),
while each piece's transfer overlaps the tracing of the next.
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


def frame_pieces(n_rows, world, chunks):
    """Cut rows [0, n_rows) into at most world*chunks tile-aligned pieces of near-equal size, dealt round-robin:
    list of (first row, rows, owner rank), in row order.  Every rank computes the same list."""
    tiles = (n_rows + TILE_ROWS - 1) // TILE_ROWS
    count = max(1, min(world * max(1, chunks), tiles))
    out = []
    for k in range(count):
        r0 = (k * tiles // count) * TILE_ROWS
        r1 = min(((k + 1) * tiles // count) * TILE_ROWS, n_rows)
        if r1 > r0:
            out.append((r0, r1 - r0, k % world))
    return out


def owned_pieces(n_rows, world, rank, chunks):
    """This rank's pieces: list of (first row, rows, first row in the rank's own buffer)."""
    out, at = [], 0
    for r0, n, owner in frame_pieces(n_rows, world, chunks):
        if owner == rank:
            out.append((r0, n, at))
            at += n
    return out


def local_rows(n_rows, world, rank, chunks):
    """Rows of the buffer a rank needs for its pieces (rank 0 needs none: it traces into the frame)."""
    return sum(n for _, n, _ in owned_pieces(n_rows, world, rank, chunks))


def trace_and_gather(trace, local, frame, n_rows, world, rank, chunks=8, group=None, stats=None):
    """One frame: trace this rank's pieces one by one and ship every finished piece to rank 0 while the next one is
    being traced.  `trace(first_row, rows, tensor)` must queue the kernel that fills `tensor` (the piece's
    [rows][cols][channels] view) on the current stream.  `local` is this rank's [local_rows][cols][channels] buffer
    (ignored on rank 0, which traces into `frame`, the whole [n_rows][cols][channels] tensor).  The transfers are
    queued behind the piece's kernel (RCCL waits for the current stream at the time of the call) and only waited
    for at the end of the frame, so the exchange of a piece can run beside the tracing of the later pieces.  RCCL's
    transfers are kernels: they only run beside a trace that leaves compute units free for them (the library's
    `reserved_cus` diagnostic; without it a transfer waits for the trace in flight to drain).  `stats`, when given,
    receives `trace_event` (first launch, last kernel done), `tail_event` (last kernel done, last transfer done),
    `wait_s` and `pieces`.  Every rank derives the same piece list, so sends and receives pair up in order."""
    if world == 1:
        trace(0, n_rows, frame)
        return
    pieces = frame_pieces(n_rows, world, chunks)
    mine = owned_pieces(n_rows, world, rank, chunks)
    works = []
    if stats is not None:
        import time
        import torch
        first, last = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        first.record()
    for c in range((len(pieces) + world - 1) // world):
        ops = []
        if c < len(mine):
            r0, n, at = mine[c]
            piece = frame[r0:r0 + n] if rank == 0 else local[at:at + n]
            trace(r0, n, piece)
            if rank != 0:
                ops.append(dist.P2POp(dist.isend, piece, 0, group))
        if rank == 0:
            for r0, n, owner in pieces[c * world + 1:(c + 1) * world]:
                ops.append(dist.P2POp(dist.irecv, frame[r0:r0 + n], owner, group))
        if ops:
            works.extend(dist.batch_isend_irecv(ops))
    if stats is not None:
        last.record()
        t0 = time.perf_counter()
    for w in works:
        w.wait()  # the current stream now waits for the transfer; the host does not
    if stats is not None:
        # `last` completes with this rank's last kernel (the transfers run on RCCL's own stream), `done` with the
        # last transfer: the time between them is the part of the exchange that no tracing on this rank hid
        done = torch.cuda.Event(enable_timing=True)
        done.record()
        stats.update(trace_event=(first, last), tail_event=(last, done), wait_s=time.perf_counter() - t0,
                     pieces=len(mine))
