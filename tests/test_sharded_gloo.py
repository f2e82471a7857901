"""world_size-2 and -4 checks of the multi-GPU sharding logic on CPU (gloo): partition / chunk ownership, the edge-frame
all_gather (also across EMPTY spans), the point-to-point overlap fetch, the limiter-maxima all-reduce and the pattern
gather (two all_gathers of plain byte tensors) + merge.
No kernel runs here; the compute side of the same decomposition is covered by
tests/test_gpu_parity.py::test_add_sharded_spans_equal_whole and ::test_sharded_stream_world1."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import audiowmark_amd as awm
from audiowmark_amd import sharded

PAY = "0123456789abcdef0011223344556677"


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _stream(total, ch):
    # deterministic "audio": every sample value identifies its global position
    return (np.arange(total * ch, dtype=np.float64) % 65521).astype(np.float32).reshape(total, ch) / 65521.0


def _worker(rank, world, port, lengths, ch, chunk_min, q, capacity=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        if capacity is not None:
            sharded.GATHER_CAPACITY = capacity      # force the second, exactly sized gather round
        awm.set_params(chunk_size_min=chunk_min)
        part = sharded.Partition(lengths)
        total = part.total
        whole = _stream(total, ch)
        s, e = part.span(rank)
        local = torch.from_numpy(whole[s:e].copy())

        # 1. edge frames for the overlap-add halo
        before, after = sharded.exchange_edge_frames(dist, local, ch)
        # the halo is the adjacent frame OF THE STREAM, whichever rank holds it (a rank in between may hold nothing)
        if s == 0:
            assert before is None
        else:
            assert np.array_equal(before.numpy(), whole[s - 1024:s])
        if e == total:
            assert after is None
        else:
            want = np.zeros((1024, ch), np.float32)
            seg = whole[e:e + 1024]
            want[:len(seg)] = seg
            assert np.array_equal(after.numpy(), want)

        # 2. overlap fetch for this rank's chunks
        buf, lo = sharded.fetch_range(dist, part, local, ch)
        g_lo, g_hi, mine = part.chunk_range(rank)
        assert lo == g_lo and buf.shape[0] == g_hi - g_lo
        assert np.array_equal(buf.numpy(), whole[g_lo:g_hi])

        # 3. limiter maxima: element-wise MAX over ranks == maxima of the whole stream
        n_blocks = total // 44100 + 2
        bm = torch.full((n_blocks,), 0.99)
        def block_maxima(values, first_sample):
            out = np.full(n_blocks, 0.99, np.float32)
            idx = (first_sample + np.arange(len(values))) // 44100
            np.maximum.at(out, idx, values)
            return out
        mine_max = block_maxima(np.abs(whole[s:e]).max(axis=1) * 2.0, s)     # pretend mixed signal, exceeds the ceiling
        bm = torch.from_numpy(mine_max.copy())
        dist.all_reduce(bm, op=dist.ReduceOp.MAX)
        assert np.array_equal(bm.numpy(), block_maxima(np.abs(whole).max(axis=1) * 2.0, 0))

        # 4. pattern gather + ResultSet merge on rank 0
        plan = part.chunk_plan()
        found = {}
        for ci, c in mine:
            # every chunk "finds" an A block 5.8 s after its start and, if long enough, the B block one block later
            pats = [dict(time=5.8, sync_index=255976, sync_quality=1.3, block_type=0, type=0, decode_error=0.1, speed=1.0, bits=PAY)]
            found[ci] = pats if ci % 2 else awm.binding.patterns_from_dicts(pats)       # both accepted forms
        merged = sharded.gather_and_merge(dist, part, None, found)
        owners = [c[3] for c in plan]
        q.put((rank, "ok", owners, None if merged is None else [round(p["time"], 3) for p in merged]))
    except Exception as exc:  # pragma: no cover
        import traceback
        q.put((rank, "fail", traceback.format_exc(), None))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("lengths,ch", [([9 * 1024 * 1000, 6_000_321], 1), ([4096 * 1000, 2_000_000], 2),
                                        ([5 * 1024 * 1000, 0, 11 * 1024 * 1000, 2_345_678], 2),      # world 4, uneven spans, one empty
                                        ([1024, 7 * 1024 * 1000, 3 * 1024 * 1000, 0], 1)])            # one frame / empty last rank
def test_sharding_over_ranks(lengths, ch):
    chunk_min = 3.0
    world = len(lengths)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    capacity = 0 if ch == 1 and world == 2 else None
    procs = [ctx.Process(target=_worker, args=(r, world, port, lengths, ch, chunk_min, q, capacity)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for r in results:
        assert r[1] == "ok", r[2]
    owners = results[0][2]
    # chunk ownership follows the chunk midpoints: non-decreasing ranks, never a rank without samples
    assert owners == sorted(owners) and all(lengths[o] > 0 for o in owners)
    awm.set_params(chunk_size_min=chunk_min)
    plan = awm.plan_chunks(sum(lengths))
    awm.set_params()
    merged = next(r[3] for r in results if r[0] == 0)
    # one pattern per chunk survives the merge (their times differ by the chunk offsets)
    assert merged == sorted(round(5.8 + c[2], 3) for c in plan)


def test_partition_rules():
    with pytest.raises(ValueError):
        sharded.Partition([1000, 2048])
    p = sharded.Partition([2048, 4096, 100])
    assert p.total == 6244 and p.span(1) == (2048, 6144) and p.owner_of(6143) == 1 and p.owner_of(6144) == 2
    # every sample a rank needs but does not own shows up in exactly one transfer
    awm.set_params(chunk_size_min=3.0)
    part = sharded.Partition([6 * 1024 * 1000, 7 * 1024 * 1000, 5_000_000])
    for r in range(3):
        lo, hi, mine = part.chunk_range(r)
        s, e = part.span(r)
        need = (hi - lo) - max(0, min(hi, e) - max(lo, s))
        got = sum(b - a for src, dst, a, b in part.transfers() if dst == r)
        assert need == got
    awm.set_params()
