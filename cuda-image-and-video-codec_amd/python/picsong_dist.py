"""Frame-sharded video encode across the GPUs of one node: one process per GPU
(torch.distributed, backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

Frames of a video are fully independent units: rank r encodes frames
r, r + N, r + 2N, ...; the only exchange is, per round of N frames, an all-gather of the codestream
lengths followed by a gatherv of the payloads to the writer rank (point-to-point sends into rank 0:
on MI355X each peer has its own xGMI link to the root, the payload is a few MB, so no ring is
needed).  Only frame 0 carries the populated 9-short header (BitStreamBuilder.cu:277-278: iter == 0),
so the rank that owns frame 0 encodes it with iter = 0 and every other frame with iter != 0.
The `_SIZE` sidecar order is frame order (IOManager.ipp:176-190).

The encoder is injected (`encode_fn(frame_index, iter) -> 1-D int16/uint16 tensor`), so the
distribution logic is testable without a GPU.
"""
import torch
import torch.distributed as dist


GATHER_CHUNK_FRAMES = 64       # gather_step: frames per grouped batch of point-to-point operations


def shard_frames(n_frames, rank, world):
    """Frame indices owned by `rank` (round-robin, f mod N)."""
    return list(range(rank, n_frames, world))


def owner_of(frame, world):
    return frame % world


def gather_round(local_stream, rank, world, device, recv_bufs=None, group=None):
    """One exchange step.  `local_stream`: this rank's codestream for the round (1-D int16 tensor on
    `device`) or None if the rank has no frame in this round.  Returns on rank 0 the list of
    per-rank streams (None for ranks without a frame), elsewhere None."""
    n = 0 if local_stream is None else int(local_stream.numel())
    mine = torch.tensor([n], dtype=torch.int32, device=device)
    lens = torch.zeros(world, dtype=torch.int32, device=device)
    dist.all_gather_into_tensor(lens, mine, group=group)
    lens = lens.tolist()
    # One grouped point-to-point batch per round: RCCL runs the 7 receives of the root as ONE kernel
    # with every peer's payload on its own xGMI link at the same time (receives issued one by one
    # would run one after the other on RCCL's stream).
    out = None
    ops = []
    if rank == 0:
        out = [local_stream if n else None]
        for r in range(1, world):
            if lens[r] == 0:
                out.append(None)
                continue
            if recv_bufs is not None:
                buf = recv_bufs[r - 1][:lens[r]]
            else:
                buf = torch.empty(lens[r], dtype=torch.int16, device=device)
            out.append(buf)
            ops.append(dist.P2POp(dist.irecv, buf, r, group))
    elif n:
        ops.append(dist.P2POp(dist.isend, local_stream, 0, group))
    if ops:
        for q in dist.batch_isend_irecv(ops):
            q.wait()
    return out


def gather_step(streams, rank, world, device, recv_bufs=None, group=None, rotate=False):
    """One exchange for a whole STEP of frames (a bucket of rounds): `streams` = this rank's codestreams of the step
    (list of 1-D int16 tensors on `device`, frame order, every rank the same count; an empty tensor = no payload).
    The lengths travel as ONE all-gather of a [world, n] tensor, the payloads as ONE grouped point-to-point batch:
    every frame's stream goes out from where the encoder put it (no packing copy) and lands back to back in the
    root's buffer for its rank, every peer on its own xGMI link at the same time.  Against gather_round per call
    this is one host wait and two collectives per step instead of three waits and two collectives per frame -- at
    5000 frames per second and GPU the per-call form is bound by launch latency, not by the links.  Returns on rank
    0 a list of `world` lists of per-frame views, elsewhere None.

    `rotate`: the writer role rotates frame by frame -- frame f of every rank's step goes to rank f mod world (and
    stays where it is on that rank).  One destination cannot take the streams of eight encoders: an 8K lossless frame
    is 20 MB every 0.2 ms, 99 GB/s per GPU, against the ~77 GB/s one direction of one xGMI link carries and the
    7 x 77 GB/s a GPU can take in at all; rotated, a link carries 1/world of a step.  Returns on EVERY rank a list of
    `world` lists of n entries: the view of frame f from rank r where f mod world == rank, None elsewhere;
    `recv_bufs`: world - 1 buffers (peers in rank order, this rank left out), each big enough for the
    ceil(n / world) frames a peer sends here, every length rounded up to a multiple of 8 shorts."""
    n = len(streams)
    mine = torch.tensor([int(t.numel()) for t in streams], dtype=torch.int32, device=device)
    lens = torch.zeros(world * n, dtype=torch.int32, device=device)
    dist.all_gather_into_tensor(lens, mine, group=group)
    lens = lens.view(world, n).tolist()
    # The point-to-point operations go out in groups of at most `chunk` frames (every rank cuts its step at the same
    # frame indices, so the sends and receives of a group match): a step of the bench is 288 frames, and one
    # batch_isend_irecv of several hundred operations per peer is more than a grouped launch needs to be.
    chunk = GATHER_CHUNK_FRAMES
    out = None
    if rotate:
        out = [[None] * n for _ in range(world)]
        bufs, offs = {}, {}
        for r in range(world):                               # what lands here: frames rank, rank + world, ... of every rank
            if r == rank:
                for f in range(rank, n, world):
                    out[r][f] = streams[f]
                continue
            # (every received stream starts on a 16-byte boundary of its buffer: RCCL's copies vectorise)
            tot = sum((lens[r][f] + 7) & ~7 for f in range(rank, n, world))
            b = r if r < rank else r - 1
            bufs[r] = recv_bufs[b][:tot] if recv_bufs is not None else torch.empty(tot, dtype=torch.int16, device=device)
            offs[r] = 0
        for c0 in range(0, n, chunk):
            ops = []
            for r in range(world):
                if r == rank:
                    continue
                for f in range(c0, min(c0 + chunk, n)):
                    if f % world != rank:
                        continue
                    v = bufs[r][offs[r]:offs[r] + lens[r][f]]
                    offs[r] += (lens[r][f] + 7) & ~7
                    out[r][f] = v
                    if lens[r][f]:
                        ops.append(dist.P2POp(dist.irecv, v, r, group))
            for f in range(c0, min(c0 + chunk, n)):          # what leaves: in frame order per destination, as it is received
                t = streams[f]
                if f % world != rank and t.numel():
                    ops.append(dist.P2POp(dist.isend, t, f % world, group))
            if ops:
                for q in dist.batch_isend_irecv(ops):
                    q.wait()
        return out
    views = None
    if rank == 0:
        out = [list(streams)]
        views = []
        for r in range(1, world):
            tot = sum(lens[r])
            buf = recv_bufs[r - 1][:tot] if recv_bufs is not None else torch.empty(tot, dtype=torch.int16, device=device)
            vr, o = [], 0
            for ln in lens[r]:
                vr.append(buf[o:o + ln])
                o += ln
            out.append(vr)
            views.append(vr)
    for c0 in range(0, n, chunk):
        ops = []
        if rank == 0:
            for r in range(1, world):
                for f in range(c0, min(c0 + chunk, n)):
                    if lens[r][f]:
                        ops.append(dist.P2POp(dist.irecv, views[r - 1][f], r, group))
        else:
            for f in range(c0, min(c0 + chunk, n)):
                if streams[f].numel():
                    ops.append(dist.P2POp(dist.isend, streams[f], 0, group))
        if ops:
            for q in dist.batch_isend_irecv(ops):
                q.wait()
    return out


class DeferredExchange:
    """Software-pipelines the per-round exchange by one frame.

    The exchange needs the codestream length on the host (send / recv counts), i.e. a wait for the
    frame.  Done right after launching the frame it would stall the launch of the next one; so the
    exchange of round i is submitted as a callable and RUN when round i + 1 has been launched (or at
    `flush`).  The GPU then always has the next frame queued while the host waits for the previous
    one, and the payload moves over xGMI while the next frame is being coded.  Results come back
    one round late, in order."""

    def __init__(self):
        self._pending = None

    def submit(self, exchange_fn):
        """`exchange_fn()` performs one round's exchange (e.g. a `gather_round` call) and returns
        its result.  Returns the result of the PREVIOUS submission (None for the first)."""
        prev, self._pending = self._pending, exchange_fn
        return prev() if prev is not None else None

    def flush(self):
        prev, self._pending = self._pending, None
        return prev() if prev is not None else None


def encode_video_distributed(n_frames, encode_fn, rank, world, device, on_frame=None, group=None,
                             pipelined=False):
    """Encode `n_frames` frames sharded f mod world; rank 0 receives every codestream in frame
    order and hands it to `on_frame(frame_index, stream_tensor)`.  Returns the per-frame lengths
    (shorts) on rank 0, None elsewhere.  `pipelined`: see DeferredExchange (then `encode_fn`
    launches the frame and returns a zero-argument callable that waits for and returns its stream)."""
    sizes = []
    rounds = (n_frames + world - 1) // world

    def deliver(rd, got):
        if rank != 0 or got is None:
            return
        for r, s in enumerate(got):
            fr = rd * world + r
            if fr >= n_frames:
                continue
            assert s is not None, f"missing stream for frame {fr}"
            sizes.append(int(s.numel()))
            if on_frame is not None:
                on_frame(fr, s)

    if not pipelined:
        for rd in range(rounds):
            f = rd * world + rank
            local = encode_fn(f, 0 if f == 0 else 1) if f < n_frames else None
            deliver(rd, gather_round(local, rank, world, device, group=group))
        return sizes if rank == 0 else None

    # pipelined: `encode_fn` only LAUNCHES the frame and returns a callable that yields its stream
    # (waiting for it); the exchange of round rd runs after round rd + 1 has been launched
    dx = DeferredExchange()
    for rd in range(rounds):
        f = rd * world + rank
        fetch = encode_fn(f, 0 if f == 0 else 1) if f < n_frames else None

        def exchange(rd=rd, fetch=fetch):
            local = fetch() if fetch is not None else None
            return rd, gather_round(local, rank, world, device, group=group)
        prev = dx.submit(exchange)
        if prev is not None:
            deliver(*prev)
    prev = dx.flush()
    if prev is not None:
        deliver(*prev)
    return sizes if rank == 0 else None


def size_sidecar(sizes):
    """`<o>_SIZE` contents: decimal lengths in shorts, comma-separated, no trailing newline."""
    return ",".join(str(int(s)) for s in sizes)


# ---- intra-frame sharding (BASELINE config 5): stripes of codeblocks of ONE frame ---------------
def stripe_ranges(n_cb, world):
    """Contiguous codeblock ranges [(begin, count)] per rank, sizes differing by at most 1."""
    q, r = divmod(n_cb, world)
    out, b = [], 0
    for k in range(world):
        n = q + (1 if k < r else 0)
        out.append((b, n))
        b += n
    return out


def splice_stripes(header9, mini_streams, counts):
    """Root side: 1-GPU codestream from the per-rank mini-streams (each 9 x 0xFFFF | count pairs |
    payload | 0xFFFF): header + all pair tables + all payloads + one trailing 0xFFFF."""
    pairs = [m[9:9 + 2 * n] for m, n in zip(mini_streams, counts)]
    payloads = [m[9 + 2 * n:-1] for m, n in zip(mini_streams, counts)]
    tail = mini_streams[0][-1:]
    return torch.cat([header9] + pairs + payloads + [tail])


def encode_frame_striped(n_cb, encode_stripe_fn, header9, rank, world, device, group=None):
    """Every rank codes its stripe (`encode_stripe_fn(begin, count) -> mini-stream tensor`); rank 0
    returns the spliced full codestream, other ranks None.  One gatherv = the only exchange."""
    ranges = stripe_ranges(n_cb, world)
    b, n = ranges[rank]
    local = encode_stripe_fn(b, n) if n > 0 else None
    got = gather_round(local, rank, world, device, group=group)
    if rank != 0:
        return None
    minis = [g for g, (_, cnt) in zip(got, ranges) if cnt > 0]
    counts = [cnt for _, cnt in ranges if cnt > 0]
    return splice_stripes(header9, minis, counts)


# ---- intra-frame sharding with a row-band transform (second form) ------------------
def band_plan(aw, ah, world):
    """Rank k's share of ONE frame when the transform is sharded too: input rows [row0, row0 + rows), the
    LL1 rows it produces (a contiguous chunk of the packed LL1 plane), and the two codeblock ranges made of
    exactly the coefficients its own band and the shared levels >= 1 produce: Mallat codeblock rows
    [k R, (k+1) R) (LL pyramid | HL1) and [AH/128 + k R, ...) (LH1 | HH1).  None when AH is not a multiple
    of 128 * world (use the stripes over a replicated transform then)."""
    if ah % (128 * world) != 0:
        return None
    rows = ah // world
    R = ah // (128 * world)
    ncx = aw // 64
    plan = []
    for k in range(world):
        plan.append({"row0": k * rows, "rows": rows,
                     "ll1_begin": (aw // 2) * (k * rows // 2), "ll1_count": (aw // 2) * (rows // 2),
                     "stripes": [(k * R * ncx, R * ncx), ((ah // 128 + k * R) * ncx, R * ncx)]})
    return plan


def encode_frame_banded(aw, ah, ops, header9, rank, world, device, group=None):
    """One frame over `world` ranks, transform included: every rank runs level 0 on its row band only
    (`ops.dwt_band(row0, rows)`), the ranks all-gather their LL1 row bands in place (`ops.ll1()` = the packed
    LL1 plane as a 1-D tensor view: the one collective of the transform, P/4 samples), run levels >= 1
    redundantly (`ops.dwt_tail()`), code their two codeblock stripes (`ops.encode_stripe(begin, count)` ->
    mini-stream) and rank 0 splices header + pair tables + payloads in raster order.  Rank 0 returns the full
    codestream -- byte-identical to the 1-GPU one -- other ranks None."""
    plan = band_plan(aw, ah, world)
    assert plan is not None, "AH must be a multiple of 128 * world for the banded transform"
    me = plan[rank]
    ops.dwt_band(me["row0"], me["rows"])
    ll1 = ops.ll1()
    if world > 1:
        mine = ll1[me["ll1_begin"]:me["ll1_begin"] + me["ll1_count"]].clone()
        dist.all_gather_into_tensor(ll1, mine, group=group)
    ops.dwt_tail()
    minis = [ops.encode_stripe(b, n) for b, n in me["stripes"]]
    # two gathers, one per half of the Mallat image, so that the root receives the stripes in raster order
    got = [gather_round(m, rank, world, device, group=group) for m in minis]
    if rank != 0:
        return None
    ordered = got[0] + got[1]
    counts = [p["stripes"][0][1] for p in plan] + [p["stripes"][1][1] for p in plan]
    return splice_stripes(header9, ordered, counts)
