"""world_size-2 checks of the multi-GPU sharding logic on CPU (gloo): partition / chunk ownership, the edge-frame
all_gather, the point-to-point overlap fetch, the limiter-maxima all-reduce and the pattern gather + merge.
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


def _worker(rank, world, port, lengths, ch, chunk_min, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        awm.set_params(chunk_size_min=chunk_min)
        part = sharded.Partition(lengths)
        total = part.total
        whole = _stream(total, ch)
        s, e = part.span(rank)
        local = torch.from_numpy(whole[s:e].copy())

        # 1. edge frames for the overlap-add halo
        before, after = sharded.exchange_edge_frames(dist, local, ch)
        if rank == 0:
            assert before is None
        else:
            assert np.array_equal(before.numpy(), whole[s - 1024:s])
        if rank == world - 1:
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
            found[ci] = pats
        merged = sharded.gather_and_merge(dist, part, None, found)
        owners = [c[3] for c in plan]
        q.put((rank, "ok", owners, None if merged is None else [round(p["time"], 3) for p in merged]))
    except Exception as exc:  # pragma: no cover
        import traceback
        q.put((rank, "fail", traceback.format_exc(), None))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("lengths,ch", [([9 * 1024 * 1000, 6_000_321], 1), ([4096 * 1000, 2_000_000], 2)])
def test_two_rank_sharding(lengths, ch):
    chunk_min = 3.0
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, lengths, ch, chunk_min, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for r in results:
        assert r[1] == "ok", r[2]
    owners = results[0][2]
    # chunk ownership follows the chunk midpoints: non-decreasing ranks, both ranks used when the stream is long enough
    assert owners == sorted(owners)
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
