"""Long streams sharded over the GPUs of one node -- one process per GPU, torch.distributed (RCCL over xGMI) as transport.

The stream is the concatenation of the ranks' local spans (rank order).  The protocol itself -- what is computed where and what
travels -- lives behind the C ABI (include/awm_hip.h: awm_sharded_add_d / awm_sharded_get_d, host/wmshard.cc):

  add   every span needs the one 1024-sample frame before and after it (3-frame overlap-add, reference wmadd.cc:228-238) and the
        limiter needs max|x| of the seconds straddling span edges (limiter.cc:99-124)
        -> one exchange of edge frames (8 KB per neighbour) + one max-reduction of the per-second maxima.
  get   the reference's chunks (wavchunkloader.cc:54-163) stay the semantic unit, the WORK of a chunk is split by position: a rank
        scores, refines and reads the blocks of the candidate starts inside its span, for which it fetches one block + 2 frames of
        its successor's samples (the "overlap stitch", 18 MB for stereo); the ranks sharing a chunk exchange its raw scores (the
        "score gather", 32 B per frame of audio), the refined scores and the blocks' soft bits; rank 0 merges the patterns.

This module supplies the transport (`TorchComm`: the three callbacks of awm_comm on top of torch.distributed point-to-point
and all_reduce) and the convenience class `ShardedStream`.  With the nccl (RCCL) backend device buffers travel GPU to GPU; with gloo
(CPU tests, and the two-process test on a single-GPU box) they are staged through host memory.
"""
import ctypes as C
from dataclasses import dataclass
from typing import List

import numpy as np

from . import binding as awm

FRAME = 1024
LIMITER_BLOCK = 44100

_EXCHANGE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_int),
                        C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_int))
_REDUCE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t)


class AwmComm(C.Structure):
    """awm_comm (include/awm_hip.h)"""
    _fields_ = [("user", C.c_void_p), ("rank", C.c_int), ("world", C.c_int), ("exchange_d", _EXCHANGE), ("exchange_h", _EXCHANGE),
                ("all_reduce_max_u32_d", _REDUCE)]


awm.lib.awm_sharded_add_d.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.POINTER(AwmComm)]
awm.lib.awm_sharded_get_d.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.POINTER(AwmComm), C.c_size_t, C.c_void_p]
awm.lib.awm_sharded_plan.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
awm.lib.awm_multi_add_d.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_char_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
awm.lib.awm_multi_get_d.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
awm.lib.awm_multi_add_watermark_batch_d.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t, C.c_void_p,
                                                    C.c_void_p, C.c_void_p, C.c_int]
awm.lib.awm_multi_get_watermark_batch_d.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_char_p, C.c_char_p, C.c_size_t, C.c_void_p, C.c_void_p,
                                                    C.c_int, C.c_size_t, C.c_void_p, C.c_void_p]


def plan(lengths):
    """awm_sharded_plan: [(chunk, rank, first_start_frame, n_start_frames)] -- which candidate start frames of which reference
    chunk every rank works on (pure host)."""
    spans = np.asarray(lengths, np.uint64)
    n = awm.lib.awm_sharded_plan(spans.ctypes.data, len(spans), 0, None, None, None, None)
    chunk, rank = np.zeros(n, np.int32), np.zeros(n, np.int32)
    first, count = np.zeros(n, np.uint64), np.zeros(n, np.uint64)
    awm.lib.awm_sharded_plan(spans.ctypes.data, len(spans), n, chunk.ctypes.data, rank.ctypes.data, first.ctypes.data, count.ctypes.data)
    return [(int(c), int(r), int(f), int(k)) for c, r, f, k in zip(chunk, rank, first, count)]


class TorchComm:
    """The transport of the sharded entry points on top of a torch.distributed process group.

    exchange_d / exchange_h: one batch_isend_irecv per round.  nccl: device buffers as they are (xGMI), host buffers staged
    through small device tensors; gloo: host buffers as they are, device buffers staged through host memory.  Messages between
    the same pair of ranks are posted in the same order on both sides (the protocol enumerates them by chunk)."""

    def __init__(self, dist, device=None, memory="cuda", ctx=None):
        # memory="host": what the protocol calls device memory is host memory (CPU tests of the wiring, no GPU involved)
        # ctx: the awm context whose entry points will call back into this transport.  awm_comm's contract is "the received bytes are
        # visible to the CONTEXT's stream on return": with it the RCCL operations are issued ON that stream (torch.cuda.ExternalStream
        # of awm_ctx_stream), whatever torch's current stream is at that moment; without it they run on torch's current stream and a
        # device synchronisation closes every exchange (correct for any stream, but the host blocks).
        import torch
        self.dist, self.torch = dist, torch
        self.ctx = ctx
        self.host_only = memory == "host"
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.nccl = dist.get_backend() == "nccl"
        self.device = device
        self.error = None
        # what this rank has sent since the last reset_stats(): bytes and rounds, device records (samples, scores: over xGMI with nccl) and
        # host records apart, and the max-reductions (bench.py: config.shard_plan)
        self.stats = {"device_bytes_sent": 0, "device_rounds": 0, "host_bytes_sent": 0, "host_rounds": 0, "max_reductions": 0, "max_reduction_bytes": 0}
        # The protocol's host-side records (refined scores, soft bits, pattern lists: a few KB per round, six rounds per `get`) over RCCL
        # would each be staged through device memory with two blocking copies; a second group over gloo (the same ranks, loopback / the
        # launcher's rendezvous address) carries them as they are.  Every rank creates it (new_group is collective); if that fails
        # anywhere the records keep going through the staged path.
        # The group is used only if EVERY rank has it: a rank that went on alone over the staged path while the others use gloo
        # would hang the first exchange_h.  The agreement is a MIN-reduction of a success flag on the default group; a failure is
        # reported on stderr (the path still works, slower), never silent.
        self.host_group = None
        if self.nccl and not self.host_only and self.world > 1:
            group, why = None, ""
            try:
                group = dist.new_group(backend="gloo")
            except Exception as e:                                # noqa: BLE001 -- any failure means "no side group here"
                why = repr(e)
            flag = torch.tensor([1 if group is not None else 0], dtype=torch.int32, device=device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 1:
                self.host_group = group
            else:
                import sys
                print(f"audiowmark_amd.sharded: rank {self.rank}: no gloo side group on every rank ({why or 'another rank failed'}); "
                      "host records travel staged through device memory over RCCL", file=sys.stderr, flush=True)
        self._cb = (_EXCHANGE(lambda *a: self._exchange(True, *a)), _EXCHANGE(lambda *a: self._exchange(False, *a)),
                    _REDUCE(self._reduce))                       # (kept alive with the object)
        self.c = AwmComm(None, self.rank, self.world, *self._cb)

    def _view(self, ptr, nbytes, on_device):
        if on_device and not self.host_only:
            return awm._as_tensor(ptr, nbytes, self.device)
        return self.torch.frombuffer((C.c_ubyte * nbytes).from_address(ptr), dtype=self.torch.uint8)

    def _ctx_stream(self):
        """torch view of the context's HIP stream (None: unknown -- no context given, or host memory only)"""
        if self.ctx is None or self.host_only:
            return None
        h = awm.lib.awm_ctx_stream(self.ctx._h)
        # (the null stream has handle 0: torch's default stream of the device is that stream)
        return self.torch.cuda.ExternalStream(h, device=self.device) if h else self.torch.cuda.default_stream(self.device)

    def reset_stats(self):
        for k in self.stats:
            self.stats[k] = 0

    def _exchange(self, on_device, user, n_send, send, send_bytes, send_to, n_recv, recv, recv_bytes, recv_from):
        try:
            kind = "device" if on_device else "host"
            self.stats[kind + "_bytes_sent"] += sum(int(send_bytes[i]) for i in range(n_send))
            self.stats[kind + "_rounds"] += 1
            stream = self._ctx_stream()
            if stream is None:
                return self._exchange_on_current(on_device, n_send, send, send_bytes, send_to, n_recv, recv, recv_bytes, recv_from, False)
            with self.torch.cuda.stream(stream):
                return self._exchange_on_current(on_device, n_send, send, send_bytes, send_to, n_recv, recv, recv_bytes, recv_from, True)
        except Exception as e:                                   # (an exception must not travel through the C frames)
            self.error = e
            return 1

    def _exchange_on_current(self, on_device, n_send, send, send_bytes, send_to, n_recv, recv, recv_bytes, recv_from, on_ctx_stream):
        try:
            torch, dist = self.torch, self.dist
            ops, copy_back = [], []
            on_device = on_device and not self.host_only
            group = None
            if not on_device and self.host_group is not None:    # host records over the gloo side group: no staging
                group = self.host_group
            direct = group is not None or on_device == self.nccl # the group moves this kind of memory itself
            for i in range(n_send):
                if not send_bytes[i]:
                    continue
                t = self._view(send[i], send_bytes[i], on_device)
                if not direct:
                    t = t.to(self.device) if self.nccl else t.cpu()
                ops.append(dist.P2POp(dist.isend, t.contiguous(), send_to[i], group))
            for i in range(n_recv):
                if not recv_bytes[i]:
                    continue
                t = self._view(recv[i], recv_bytes[i], on_device)
                if not direct:
                    stage = torch.empty(recv_bytes[i], dtype=torch.uint8, device=self.device if self.nccl else "cpu")
                    copy_back.append((t, stage))
                    t = stage
                ops.append(dist.P2POp(dist.irecv, t, recv_from[i], group))
            if ops:
                for req in dist.batch_isend_irecv(ops):
                    req.wait()
            for t, stage in copy_back:
                t.copy_(stage)
            # awm_comm's contract: the writes are visible to the CONTEXT's stream on return.  nccl + device buffers: req.wait() has
            # made torch's CURRENT stream wait for the transfers -- that is the context's stream when the caller handed the context
            # over (on_ctx_stream), so everything stays stream ordered and the host does not block (the library orders its lanes
            # behind the context's stream itself).  Without the context nothing says the two streams are the same one: a device
            # synchronisation then closes the exchange.  Staged paths (gloo, host buffers over nccl) end in blocking copies; a
            # device sync closes those as well.
            stream_ordered = self.nccl and on_device and on_ctx_stream
            if not self.host_only and not stream_ordered and group is None:
                torch.cuda.synchronize(self.device)
            return 0
        except Exception as e:                                   # (an exception must not travel through the C frames)
            self.error = e
            return 1

    def _reduce(self, user, data, n):
        try:
            self.stats["max_reductions"] += 1
            self.stats["max_reduction_bytes"] += int(n) * 4
            torch, dist = self.torch, self.dist
            stream = self._ctx_stream()
            # non-negative floats order like their bit patterns: max of the words
            t = self._view(data, n * 4, True).view(torch.int32)
            if self.nccl or self.host_only:
                if stream is not None:
                    with torch.cuda.stream(stream):               # the reduction is ordered on the CONTEXT's stream
                        dist.all_reduce(t, op=dist.ReduceOp.MAX)
                else:
                    dist.all_reduce(t, op=dist.ReduceOp.MAX)
            else:
                h = t.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.MAX)
                t.copy_(h)
            if not self.host_only and not (self.nccl and stream is not None):   # (nccl on the context's stream: stream ordered)
                torch.cuda.synchronize(self.device)
            return 0
        except Exception as e:
            self.error = e
            return 1


@dataclass
class Partition:
    """Contiguous spans of a stream, one per rank; every span but the last non-empty one must be a whole number of frames."""
    lengths: List[int]

    def __post_init__(self):
        last = max((i for i, n in enumerate(self.lengths) if n), default=-1)
        for i, n in enumerate(self.lengths):
            if n % FRAME and i != last:
                raise ValueError("every span except the last non-empty one must be a multiple of 1024 samples")
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

    def work(self):
        """candidate start frames per rank over all chunks (awm_sharded_plan): what the time of `get` is proportional to"""
        out = [0] * len(self.lengths)
        for _, r, _, k in plan(self.lengths):
            out[r] += k
        return out


class ShardedStream:
    """add / get on a stream whose spans live on the ranks of `dist` (one GPU each)."""

    def __init__(self, ctx, dist, n_frames_local, n_channels):
        import torch
        self.ctx, self.dist, self.n_channels = ctx, dist, n_channels
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        dev = torch.device("cuda", ctx.device)
        self.comm = TorchComm(dist, dev, ctx=ctx)
        mine = torch.tensor([n_frames_local], dtype=torch.int64, device=dev if self.comm.nccl else "cpu")
        lens = [torch.zeros_like(mine) for _ in range(self.world)]
        dist.all_gather(lens, mine)
        self.part = Partition([int(t.item()) for t in lens])
        self._spans = np.asarray(self.part.lengths, np.uint64)

    def _fail(self, what):
        if self.comm.error is not None:
            err, self.comm.error = self.comm.error, None
            raise awm.AwmError(f"{what}: transport failed: {err!r}") from err

    def add_watermark(self, key, payload_hex, local, out, sample_rate=44100):
        if sample_rate != 44100:
            # spans are cut in watermark frames and limiter blocks of 44 100 samples; other rates go through the resampled
            # path of the single-GPU add (awm_add_watermark_d), which is not sharded
            raise ValueError("ShardedStream.add_watermark: only 44.1 kHz streams are sharded")
        rc = awm.lib.awm_sharded_add_d(self.ctx._h, awm.key_bytes(key), payload_hex.encode(), awm._dev_ptr(local), awm._dev_ptr(out),
                                       self.n_channels, self._spans.ctypes.data, C.byref(self.comm.c))
        self._fail("awm_sharded_add_d")
        awm._check(rc, "awm_sharded_add_d")
        return out

    def get_watermark(self, key, local, max_out=8192):
        """merged pattern list on rank 0, None elsewhere (8192 patterns: > 60 h of audio)"""
        buf = self.ctx._pattern_buffer(max_out)
        rc = awm.lib.awm_sharded_get_d(self.ctx._h, awm.key_bytes(key), awm._dev_ptr(local), self.n_channels, self._spans.ctypes.data,
                                       C.byref(self.comm.c), max_out, C.cast(buf, C.c_void_p))
        self._fail("awm_sharded_get_d")
        cnt = awm._check(rc, "awm_sharded_get_d")
        if self.rank != 0:
            return None
        if cnt > max_out:
            raise awm.AwmError("awm_sharded_get_d: more patterns than the buffer holds")
        return awm.patterns_to_dicts(buf, cnt)


def multi_add(ctxs, key, payload_hex, spans_in, spans_out):
    """awm_multi_add_d: one stream over the contexts of ONE process (span i on ctxs[i]'s device), hipMemcpyPeer as transport"""
    n = len(ctxs)
    shapes = [awm._pcm_shape(t) for t in spans_in]
    ch = shapes[0][1]
    h = (C.c_void_p * n)(*[c._h for c in ctxs])
    pin = (C.c_void_p * n)(*[awm._dev_ptr(t) for t in spans_in])
    pout = (C.c_void_p * n)(*[awm._dev_ptr(t) for t in spans_out])
    lens = np.asarray([s[0] for s in shapes], np.uint64)
    awm._check(awm.lib.awm_multi_add_d(h, n, awm.key_bytes(key), payload_hex.encode(), pin, pout, ch, lens.ctypes.data), "awm_multi_add_d")
    return spans_out


def multi_get(ctxs, key, spans, max_out=4096):
    """awm_multi_get_d: the merged pattern list of one stream whose spans live on the contexts of one process"""
    n = len(ctxs)
    shapes = [awm._pcm_shape(t) for t in spans]
    ch = shapes[0][1]
    h = (C.c_void_p * n)(*[c._h for c in ctxs])
    ptr = (C.c_void_p * n)(*[awm._dev_ptr(t) for t in spans])
    lens = np.asarray([s[0] for s in shapes], np.uint64)
    while True:
        buf = ctxs[0]._pattern_buffer(max_out)
        cnt = awm._check(awm.lib.awm_multi_get_d(h, n, awm.key_bytes(key), ptr, ch, lens.ctypes.data, max_out, C.cast(buf, C.c_void_p)),
                         "awm_multi_get_d")
        if cnt <= max_out:
            return awm.patterns_to_dicts(buf, cnt)
        max_out = cnt


def _clip_args(ctxs, keys, clips, owner):
    n = len(clips)
    shapes = [awm._pcm_shape(c) for c in clips]
    ch = shapes[0][1]
    assert all(s[1] == ch for s in shapes) and len(owner) == n
    h = (C.c_void_p * len(ctxs))(*[c._h for c in ctxs])
    own = (C.c_int * n)(*owner)
    ptrs = (C.c_void_p * n)(*[awm._dev_ptr(c) for c in clips])
    frames = (C.c_size_t * n)(*[s[0] for s in shapes])
    if isinstance(keys, (list, tuple)):
        assert len(keys) == n
        flat, one = b"".join(awm.key_bytes(k) for k in keys), None
    else:
        flat, one = None, awm.key_bytes(keys)
    return h, own, ptrs, frames, ch, flat, one


def multi_add_batch(ctxs, keys, payload_hex, clips, owner, outs=None):
    """awm_multi_add_watermark_batch_d: independent clips over the contexts of one process; clip i (resident on the device of
    ctxs[owner[i]]) is watermarked there.  keys: a list with one key per clip, or one key (or None) for all."""
    import torch
    if outs is None:
        outs = [torch.empty_like(c) for c in clips]
    h, own, src, frames, ch, flat, one = _clip_args(ctxs, keys, clips, owner)
    dst = (C.c_void_p * len(clips))(*[awm._dev_ptr(o) for o in outs])
    awm._check(awm.lib.awm_multi_add_watermark_batch_d(h, len(ctxs), own, flat, one, payload_hex.encode(), len(clips), src, dst, frames, ch),
               "awm_multi_add_watermark_batch_d")
    return outs


def multi_get_batch(ctxs, keys, clips, owner, max_out_per_clip=64):
    """awm_multi_get_watermark_batch_d: the pattern lists of independent clips, clip i decoded on ctxs[owner[i]]"""
    h, own, ptrs, frames, ch, flat, one = _clip_args(ctxs, keys, clips, owner)
    n = len(clips)
    n_out = (C.c_int * n)()
    while True:
        buf = ctxs[0]._pattern_buffer(n * max_out_per_clip)
        awm._check(awm.lib.awm_multi_get_watermark_batch_d(h, len(ctxs), own, flat, one, n, ptrs, frames, ch, max_out_per_clip,
                                                           C.cast(buf, C.c_void_p), n_out), "awm_multi_get_watermark_batch_d")
        most = max(n_out) if n else 0
        if most <= max_out_per_clip:
            return [awm.patterns_to_dicts(buf, n_out[i], i * max_out_per_clip) for i in range(n)]
        max_out_per_clip = most
