"""The N>1 plumbing (all-gather of padded sketch units + device-side CSR compaction + query-block split) on CPU:
two processes, gloo backend, 127.0.0.1.  On the GPU box the same code runs over RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import kssd_oracle as ko


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def make_rank_data(rank, G, cap):
    rng = np.random.default_rng(100 + rank)
    sizes = rng.integers(0, 40, G)
    sizes[rng.integers(0, G)] = 0
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    ids = np.zeros(cap, dtype=np.int32)
    vals = np.concatenate([np.sort(rng.choice(5000, int(s), replace=False)) for s in sizes]) if off[-1] else np.zeros(0)
    ids[:off[-1]] = vals
    ids[off[-1]:] = -7  # padding garbage must never surface
    return off, ids


def _worker(rank, world, port, G, cap, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from public_kssd_amd.shard import SketchGather, query_block
        off, ids = make_rank_data(rank, G, cap)
        g = SketchGather(world, G, cap, torch.device("cpu"))
        for _ in range(2):  # reusable without re-allocation
            roff, rids = g(torch.from_numpy(off), torch.from_numpy(ids))
        q.put((rank, roff.numpy().copy(), rids.numpy().copy(), query_block(rank, G)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_all_gather_and_compaction(world):
    G, cap = 17, 800
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, G, cap, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    want_off = [0]
    want_ids = []
    for r in range(world):
        off, ids = make_rank_data(r, G, cap)
        for g in range(G):
            want_ids.append(ids[off[g]:off[g + 1]])
            want_off.append(want_off[-1] + int(off[g + 1] - off[g]))
    want_ids = np.concatenate(want_ids)
    blocks = set()
    for rank, roff, rids, (qb, qe) in res:
        assert np.array_equal(roff, np.array(want_off))
        assert np.array_equal(rids[:roff[-1]], want_ids)
        assert (qb, qe) == (rank * G, (rank + 1) * G)
        blocks.add((qb, qe))
        # rows of the rank's block computed against the gathered references == the same rows of the global matrix
        full = ko.shared_counts(roff.astype(np.uint64), rids[:roff[-1]].astype(np.uint32), roff.astype(np.uint64),
                                rids[:roff[-1]].astype(np.uint32))
        assert np.array_equal(np.diag(full)[qb:qe], np.diff(roff)[qb:qe])
    assert sorted(blocks) == [(r * G, (r + 1) * G) for r in range(world)]   # the blocks tile the query axis
