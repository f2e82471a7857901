"""Long streams sharded over the GPUs of one node -- one process per GPU, torch.distributed (RCCL over xGMI).

The stream is the concatenation of the ranks' local spans (rank order).  Nothing here touches sample data on the
host; the only traffic between ranks is what the algorithm itself couples:

  add   every span needs the one 1024-sample frame before and after it (3-frame overlap-add, reference
        wmadd.cc:228-238) and the limiter needs max|x| of the seconds straddling span edges (limiter.cc:99-124)
        -> one all_gather of the edge frames (16 KB per rank) + one all_reduce(MAX) of the per-second maxima.
  get   the reference's own chunks (wavchunkloader.cc:54-163) are the shard unit; a chunk is decoded by the rank
        that holds its midpoint, the part of it that lives on a neighbour is fetched point-to-point (the "overlap
        stitch"), found patterns are gathered on rank 0 and merged with ResultSet semantics (wmget.cc:288-316).

`Partition` and the exchange helpers are backend agnostic (they are exercised with gloo / CPU tensors in
tests/test_sharded_gloo.py); the compute calls need a GPU.
"""
from dataclasses import dataclass
from typing import List, Tuple

import numpy as np

from . import binding as awm

FRAME = 1024
LIMITER_BLOCK = 44100


class _Comm:
    """Thin view of torch.distributed that also works when the process group cannot move device tensors itself
    (gloo with CUDA tensors: used to test the multi-rank numerics on a single-GPU box).  With the nccl (RCCL) backend
    every call goes straight to torch.distributed and the data stays on the devices (xGMI)."""

    def __init__(self, dist):
        self.dist = dist
        self.stage = dist.get_backend() != "nccl"

    def _h(self, t):
        return t.cpu() if (self.stage and t.is_cuda) else t

    def all_gather(self, outs, t):
        if not (self.stage and t.is_cuda):
            return self.dist.all_gather(outs, t)
        host = [o.cpu() for o in outs]
        self.dist.all_gather(host, t.cpu())
        for o, h in zip(outs, host):
            o.copy_(h)

    def all_reduce_max(self, t):
        if not (self.stage and t.is_cuda):
            return self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        h = t.cpu()
        self.dist.all_reduce(h, op=self.dist.ReduceOp.MAX)
        t.copy_(h)

    def exchange_start(self, sends, recvs):
        """Post the copies and return a function that completes them.  With RCCL the transfers run on the collective
        stream while the caller keeps computing; the staged (gloo) variant is synchronous."""
        if self.stage:
            self.exchange(sends, recvs)
            return lambda: None
        dist = self.dist
        ops = [dist.P2POp(dist.isend, t.contiguous(), dst) for t, dst in sends]
        ops += [dist.P2POp(dist.irecv, t, src) for t, src in recvs]
        reqs = dist.batch_isend_irecv(ops) if ops else []

        def finish():
            for r in reqs:
                r.wait()
        return finish

    def exchange(self, sends, recvs):
        """sends: [(tensor, dst)], recvs: [(tensor, src)] -- all posted together, completed before returning."""
        dist = self.dist
        host_recv = []
        ops = []
        for t, dst in sends:
            ops.append(dist.P2POp(dist.isend, self._h(t).contiguous(), dst))
        for t, src in recvs:
            h = torch_empty_like_host(t) if (self.stage and t.is_cuda) else t
            host_recv.append((t, h))
            ops.append(dist.P2POp(dist.irecv, h, src))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        for t, h in host_recv:
            if h is not t:
                t.copy_(h)


def torch_empty_like_host(t):
    import torch
    return torch.empty(t.shape, dtype=t.dtype)


@dataclass
class Partition:
    """Contiguous spans of a stream, one per rank; every span but the last must be a whole number of frames."""
    lengths: List[int]

    def __post_init__(self):
        for n in self.lengths[:-1]:
            if n % FRAME:
                raise ValueError("every span except the last one must be a multiple of 1024 samples")
        self.starts = [0]
        for n in self.lengths:
            self.starts.append(self.starts[-1] + n)

    @property
    def total(self):
        return self.starts[-1]

    def span(self, rank):
        return self.starts[rank], self.starts[rank + 1]

    def owner_of(self, sample):
        for r in range(len(self.lengths)):
            if self.starts[r] <= sample < self.starts[r + 1]:
                return r
        return len(self.lengths) - 1

    # ---- get: reference chunks ----------------------------------------------------------------
    def chunk_plan(self):
        """[(first_frame, n_frames, time_offset, owner_rank)] for the whole stream (computed once: it is consulted several
        times per `get`, inside the timed region)."""
        plan = getattr(self, "_plan", None)
        if plan is None:
            plan = []
            for first, count, off in awm.plan_chunks(self.total):
                plan.append((first, count, off, self.owner_of(first + count // 2)))
            self._plan = plan
        return plan

    def chunk_range(self, rank):
        """(lo, hi, chunks) -- the global sample range rank must hold to decode its chunks; chunks are consecutive."""
        mine = [(i, c) for i, c in enumerate(self.chunk_plan()) if c[3] == rank]
        if not mine:
            return 0, 0, []
        lo = min(c[0] for _, c in mine)
        hi = max(c[0] + c[1] for _, c in mine)
        return lo, hi, mine

    def transfers(self):
        """All point-to-point copies needed before `get`: (src_rank, dst_rank, global_lo, global_hi)."""
        cached = getattr(self, "_transfers", None)
        if cached is not None:
            return cached
        out = []
        for dst in range(len(self.lengths)):
            lo, hi, _ = self.chunk_range(dst)
            for src in range(len(self.lengths)):
                if src == dst:
                    continue
                s, e = self.span(src)
                a, b = max(lo, s), min(hi, e)
                if a < b:
                    out.append((src, dst, a, b))
        self._transfers = out
        return out


def exchange_edge_frames(dist, local, n_channels, lengths=None):
    """all_gather of each rank's first and last frame -> (halo_before, halo_after) for this rank (None at the ends).
    A last frame shorter than 1024 samples is zero padded (it can only be the end of the stream).  Ranks with an EMPTY span
    are skipped: the halo is the adjacent frame of the stream, i.e. the edge frame of the nearest rank that holds samples.
    `lengths` = the span lengths of all ranks (Partition.lengths) if the caller knows them, else they are gathered."""
    import torch
    rank, world = dist.get_rank(), dist.get_world_size()
    n = local.shape[0]
    comm = _Comm(dist)
    if lengths is None:
        lens = [torch.zeros(1, dtype=torch.int64, device=local.device) for _ in range(world)]
        comm.all_gather(lens, torch.tensor([n], dtype=torch.int64, device=local.device))
        lengths = [int(t.item()) for t in lens]
    edge = torch.zeros((2, FRAME, n_channels), dtype=local.dtype, device=local.device)
    view = local.reshape(n, n_channels)
    edge[0, :min(FRAME, n)] = view[:FRAME]
    last_start = max(0, ((n - 1) // FRAME) * FRAME) if n else 0
    tail = view[last_start:]
    edge[1, :tail.shape[0]] = tail
    gathered = [torch.empty_like(edge) for _ in range(world)]
    comm.all_gather(gathered, edge)
    prev = next((r for r in range(rank - 1, -1, -1) if lengths[r]), None)
    nxt = next((r for r in range(rank + 1, world) if lengths[r]), None)
    before = gathered[prev][1].contiguous() if prev is not None else None
    after = gathered[nxt][0].contiguous() if nxt is not None else None
    return before, after


def fetch_range(dist, part: Partition, local, n_channels):
    """Assemble the global sample range this rank needs for its chunks: own samples are copied, the rest arrives
    point-to-point from the neighbours.  Returns (buffer, lo)."""
    import torch
    rank = dist.get_rank()
    lo, hi, _ = part.chunk_range(rank)
    my_s, my_e = part.span(rank)
    view = local.reshape(local.shape[0], n_channels)
    buf = torch.empty((max(0, hi - lo), n_channels), dtype=local.dtype, device=local.device)
    a, b = max(lo, my_s), min(hi, my_e)
    if a < b:
        buf[a - lo:b - lo] = view[a - my_s:b - my_s]
    sends, recvs = [], []
    for src, dst, g_lo, g_hi in part.transfers():
        if src == rank:
            sends.append((view[g_lo - my_s:g_hi - my_s].contiguous(), dst))
        elif dst == rank:
            recvs.append((torch.empty((g_hi - g_lo, n_channels), dtype=local.dtype, device=local.device), src, g_lo))
    _Comm(dist).exchange(sends, [(t, src) for t, src, _ in recvs])
    for t, _, g_lo in recvs:
        buf[g_lo - lo:g_lo - lo + t.shape[0]] = t
    return buf, lo


GATHER_CAPACITY = 512          # records per rank in the one-shot gather (an hour of audio yields ~110 patterns)


def gather_patterns(dist, my_chunk_patterns):
    """{chunk index: structured array (PATTERN_DTYPE)} of this rank -> the union over all ranks, on every rank.
    ONE collective on plain byte tensors (the "score gather" of the path): every rank contributes a fixed-size block = its record
    count followed by GATHER_CAPACITY records (chunk index int32 + pattern); only if some rank found more than that, a second
    all_gather sized for the longest list follows."""
    import torch
    comm = _Comm(dist)
    world = dist.get_world_size()
    rec = np.dtype([("chunk", np.int32), ("pattern", awm.PATTERN_DTYPE)])
    mine = np.zeros(sum(len(p) for p in my_chunk_patterns.values()), rec)
    pos = 0
    for ci in sorted(my_chunk_patterns):
        pats = my_chunk_patterns[ci]
        mine["chunk"][pos:pos + len(pats)] = ci
        mine["pattern"][pos:pos + len(pats)] = pats
        pos += len(pats)
    dev = getattr(dist, "_awm_device", None) or "cpu"

    def gather(capacity):
        block = np.zeros(16 + capacity * rec.itemsize, np.uint8)
        block[:8] = np.frombuffer(np.int64(len(mine)).tobytes(), np.uint8)
        k = min(len(mine), capacity)
        block[16:16 + k * rec.itemsize] = mine[:k].view(np.uint8).reshape(-1)
        t = torch.from_numpy(block).to(dev)
        gathered = torch.empty((world, t.numel()), dtype=t.dtype, device=t.device)     # one buffer, one copy back
        comm.all_gather(list(gathered.unbind(0)), t)
        host = gathered.cpu().numpy()
        blocks = [host[r] for r in range(world)]
        counts = [int(np.frombuffer(b[:8].tobytes(), np.int64)[0]) for b in blocks]
        return blocks, counts

    blocks, counts = gather(GATHER_CAPACITY)
    if max(counts) > GATHER_CAPACITY:
        blocks, counts = gather(max(counts))
    out = {}
    for b, n in zip(blocks, counts):
        recs = np.frombuffer(b[16:16 + n * rec.itemsize].tobytes(), rec)
        for ci in np.unique(recs["chunk"]):
            out[int(ci)] = recs["pattern"][recs["chunk"] == ci].copy()
    return out


def gather_and_merge(dist, part: Partition, key, my_chunk_patterns):
    """my_chunk_patterns: {global chunk index: patterns with chunk relative times} -> merged list on rank 0 (None elsewhere).
    The values are structured arrays (binding.PATTERN_DTYPE) or lists of pattern dicts (converted)."""
    plan = part.chunk_plan()
    payload = {}
    for ci, pats in my_chunk_patterns.items():
        if not isinstance(pats, np.ndarray):
            pats = awm.patterns_from_dicts(pats)
        pats = pats.copy()
        pats["time"] += plan[ci][2]                                            # ResultSet::apply_time_offset
        payload[ci] = pats
    everything = gather_patterns(dist, payload)
    if dist.get_rank() != 0:
        return None
    per_chunk = [everything.get(ci, np.zeros(0, awm.PATTERN_DTYPE)) for ci in range(len(plan))]
    return awm.merge_patterns_raw(key, per_chunk)


class ShardedStream:
    """add / get on a stream whose spans live on the ranks of `dist` (one GPU each)."""

    def __init__(self, ctx, dist, n_frames_local, n_channels):
        import torch
        self.ctx, self.dist, self.n_channels = ctx, dist, n_channels
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        lens = [torch.zeros(1, dtype=torch.int64, device=self._device()) for _ in range(self.world)]
        _Comm(dist).all_gather(lens, torch.tensor([n_frames_local], dtype=torch.int64, device=self._device()))
        self.part = Partition([int(t.item()) for t in lens])
        self._fm_cache = {}
        dist._awm_device = self._device() if dist.get_backend() == "nccl" else "cpu"     # where the small collectives live

    def _device(self):
        import torch
        return torch.device("cuda", self.ctx.device)

    def _frame_mod(self, key, payload_hex):
        k = (awm.key_bytes(key), payload_hex)
        if k not in self._fm_cache:
            self._fm_cache[k] = awm.tab_frame_mod(key, payload_hex)
        return self._fm_cache[k]

    def add_watermark(self, key, payload_hex, local, out, water_delta=0.01, use_limiter=True, sample_rate=44100):
        import torch
        if sample_rate != 44100:
            # spans are cut in watermark frames and limiter blocks of 44 100 samples; other rates go through the resampled
            # path of the single-GPU add (awm_add_watermark_d), which is not sharded
            raise ValueError("ShardedStream.add_watermark: only 44.1 kHz streams are sharded")
        dist = self.dist
        start, _ = self.part.span(self.rank)
        before, after = exchange_edge_frames(dist, local, self.n_channels, self.part.lengths)
        block_max = None
        if use_limiter:
            n_blocks = self.part.total // LIMITER_BLOCK + 2
            block_max = torch.empty(n_blocks, dtype=torch.float32, device=local.device)
            self.ctx.add_init_block_max(block_max)
        self.ctx.add_mix(local, out, self._frame_mod(key, payload_hex), water_delta, start // FRAME, before, after, block_max)
        if use_limiter:
            _Comm(dist).all_reduce_max(block_max)                  # seconds that straddle span edges
            self.ctx.add_limit(out, start, block_max)
        return out

    def get_watermark(self, key, local):
        """Chunks that lie completely inside this rank's span are decoded while the parts of the straddling chunks that
        live on the neighbours are still in flight (up to half a chunk = 300 MB per boundary for 30 minute chunks: several
        milliseconds over one xGMI link, about as long as decoding a chunk)."""
        import torch
        part, rank, C = self.part, self.rank, self.n_channels
        lo, hi, mine = part.chunk_range(rank)
        my_s, my_e = part.span(rank)
        view = local.reshape(local.shape[0], C)
        inside = [(ci, c) for ci, c in mine if c[0] >= my_s and c[0] + c[1] <= my_e]
        cross = [(ci, c) for ci, c in mine if not (c[0] >= my_s and c[0] + c[1] <= my_e)]
        # post the transfers (every rank serves its neighbours even if it needs nothing itself)
        sends, recvs = [], []
        for src, dst, g_lo, g_hi in part.transfers():
            if src == rank:
                sends.append((view[g_lo - my_s:g_hi - my_s], dst))
            elif dst == rank:
                recvs.append((torch.empty((g_hi - g_lo, C), dtype=local.dtype, device=local.device), src, g_lo))
        finish = _Comm(self.dist).exchange_start(sends, [(t, src) for t, src, _ in recvs])
        found = {}
        if inside:
            rel = [(c[0] - my_s, c[1]) for _, c in inside]
            pats, which = self.ctx.decode_chunks_raw(key, view, rel, first_is_stream_start=(inside[0][1][0] == 0))
            found.update({ci: pats[which == i] for i, (ci, _) in enumerate(inside)})
        finish()
        if cross:
            c_lo = min(c[0] for _, c in cross)
            c_hi = max(c[0] + c[1] for _, c in cross)
            buf = torch.empty((c_hi - c_lo, C), dtype=local.dtype, device=local.device)
            a, b = max(c_lo, my_s), min(c_hi, my_e)
            if a < b:
                buf[a - c_lo:b - c_lo] = view[a - my_s:b - my_s]
            for t, _, g_lo in recvs:
                a, b = max(c_lo, g_lo), min(c_hi, g_lo + t.shape[0])
                if a < b:
                    buf[a - c_lo:b - c_lo] = t[a - g_lo:b - g_lo]
            rel = [(c[0] - c_lo, c[1]) for _, c in cross]
            pats, which = self.ctx.decode_chunks_raw(key, buf, rel, first_is_stream_start=(cross[0][1][0] == 0))
            found.update({ci: pats[which == i] for i, (ci, _) in enumerate(cross)})
        return gather_and_merge(self.dist, self.part, key, found)
