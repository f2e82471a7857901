"""Multi-GPU path on CPU: the work plan of the sharded `get` (awm_sharded_plan: coverage, balance, uneven / empty spans) and the
torch.distributed transport (audiowmark_amd.sharded.TorchComm) behind the C ABI's awm_comm callbacks, with gloo at world size 2
and 4 -- every kind of message the protocol sends (several per pair and round, empty ones, device-kind and host-kind buffers,
the max-reduction), called through the C function pointers exactly like host/wmshard.cc calls them.
No kernel runs here; the compute side is covered on the GPU by tests/test_gpu_parity.py (awm_multi_* with several contexts on one
device == the single-GPU result; two processes over gloo == the single-process result)."""
import ctypes as C
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import audiowmark_amd as awm
from audiowmark_amd import sharded

BLOCK = 2226


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _payload(src, dst, k, nbytes):
    return ((np.arange(nbytes, dtype=np.int64) * 131 + src * 7919 + dst * 104729 + k * 1299709) % 251).astype(np.uint8)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        comm = sharded.TorchComm(dist, memory="host")
        c = comm.c
        assert (c.rank, c.world) == (rank, world)
        for kind in ("exchange_d", "exchange_h"):
            fn = getattr(c, kind)
            # every rank sends every other rank three messages (sizes depend on the pair; the middle one is empty)
            sends, recvs, expect = [], [], []
            for peer in range(world):
                if peer == rank:
                    continue
                for k, nbytes in enumerate((1000 + 17 * rank + 5 * peer, 0, 70001 + rank)):
                    sends.append((_payload(rank, peer, k, nbytes), peer))
                for k, nbytes in enumerate((1000 + 17 * peer + 5 * rank, 0, 70001 + peer)):
                    recvs.append((np.zeros(nbytes, np.uint8), peer))
                    expect.append(_payload(peer, rank, k, nbytes))
            ptr = lambda arrs: (C.c_void_p * len(arrs))(*[a.ctypes.data if a.size else None for a, _ in arrs])
            size = lambda arrs: (C.c_size_t * len(arrs))(*[a.size for a, _ in arrs])
            peers = lambda arrs: (C.c_int * len(arrs))(*[p for _, p in arrs])
            rc = fn(None, len(sends), ptr(sends), size(sends), peers(sends), len(recvs), ptr(recvs), size(recvs), peers(recvs))
            assert rc == 0, comm.error
            for (got, _), want in zip(recvs, expect):
                assert np.array_equal(got, want)
            # a round in which this rank has nothing to do still returns
            assert fn(None, 0, None, None, None, 0, None, None, None) == 0
        # limiter maxima: element-wise maximum of non-negative floats through their bit patterns
        vals = np.abs(np.sin(np.arange(4097, dtype=np.float32) * (rank + 1))).astype(np.float32)
        buf = vals.copy()
        assert c.all_reduce_max_u32_d(None, buf.ctypes.data, buf.size) == 0, comm.error
        want = np.max([np.abs(np.sin(np.arange(4097, dtype=np.float32) * (r + 1))).astype(np.float32) for r in range(world)], axis=0)
        assert np.array_equal(buf, want)
        # a transport failure is reported through the return code, the exception is kept for the caller
        eight = np.zeros(8, np.uint8)
        one_ptr, one_size, one_peer = (C.c_void_p * 1)(eight.ctypes.data), (C.c_size_t * 1)(8), (C.c_int * 1)((rank + 1) % world)
        group, comm.dist = comm.dist, None
        assert c.exchange_h(None, 1, one_ptr, one_size, one_peer, 0, None, None, None) == 1 and comm.error is not None
        comm.dist = group
        q.put((rank, "ok"))
    except Exception as e:                                       # pragma: no cover
        import traceback
        q.put((rank, "FAILED: " + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_torch_transport_over_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in results), results


def test_partition_rules():
    with pytest.raises(ValueError):
        sharded.Partition([1000, 2048])
    p = sharded.Partition([2048, 4096, 100])
    assert p.total == 6244 and p.span(1) == (2048, 6144) and p.owner_of(6143) == 1 and p.owner_of(6144) == 2
    # a ragged tail may be followed by ranks that hold nothing
    p = sharded.Partition([4096, 2_345_678, 0, 0])
    assert p.total == 4096 + 2_345_678
    with pytest.raises(ValueError):
        sharded.Partition([4096, 2_345_678, 0, 1024])


def _check_plan(lengths):
    """every candidate start frame of every chunk belongs to exactly one rank, ranks in position order"""
    total = sum(lengths)
    chunks = awm.plan_chunks(total)
    entries = sharded.plan(lengths)
    assert len(entries) == len(chunks) * len(lengths)
    work = [0] * len(lengths)
    for ci, (first, count, _) in enumerate(chunks):
        S = count // 1024 - 1 - BLOCK
        mine = [e for e in entries if e[0] == ci]
        pos = 0
        for _, r, lo, n in mine:
            if n:
                assert lo == pos, (ci, r, lo, pos)
                # the rank holds the sample one frame before the first start frame it works on
                g = first + max(lo - 1, 0) * 1024
                s = sum(lengths[:r])
                assert s <= g < s + lengths[r], (ci, r, g, s)
                pos += n
                work[r] += n
        assert pos == max(S, 0), (ci, pos, S)
    return work


def test_plan_covers_every_start_frame_and_balances():
    rate = 44100
    # BASELINE configs[3]: 8 h over 8 ranks -- 18 reference chunks.  Every rank works on what lies in its span; what differs from
    # rank to rank is how many of the 17 chunk overlaps (134 s = 2.6 blocks each, decoded twice by the reference's design) fall
    # into it: two or three.  Balanced to within one overlap: 2 % over the mean.
    per = 3600 * rate // 1024 * 1024
    overlap = (2 * 1.3 * BLOCK)
    work = _check_plan([per] * 7 + [8 * 3600 * rate - 7 * per])
    assert max(work) - min(work) <= overlap + 2 * BLOCK, work
    assert max(work) / (sum(work) / 8) < 1.03
    # (ownership of whole chunks, the round 2 design, put 3 of 18 chunks on some ranks: 1.33 x the mean)
    # weak scaling shape of bench.py: N x 60 min
    for n in (2, 4):
        work = _check_plan([per] * n)
        assert max(work) - min(work) <= overlap + 2 * BLOCK, work
    # uneven spans, an empty one, spans shorter than a block (their start frames need samples of several successors), a ragged tail
    _check_plan([5 * per // 4 // 1024 * 1024, 0, 1024 * 1000, 1024 * 50, 3 * per // 4 // 1024 * 1024 + 333])
    _check_plan([1024 * 7000, 1024 * 3, 0, 1024 * 9000 + 17])
    _check_plan([rate * 400 // 1024 * 1024])            # one rank: all its own
    assert all(n == 0 for _, _, _, n in sharded.plan([1024 * 100, 1024 * 100]))      # too short for the block decoder: no parts
