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


class OracleEngine:
    """the two device-level entry points ShardedSearch needs, computed by the CPU oracle on CPU tensors (test
    infrastructure: lets the sharding logic run under gloo without a GPU)"""

    def index_build_device(self, roff, rids, n_ref, max_ids, stream=None, check=True):
        self.roff = roff.numpy()[:n_ref + 1].astype(np.uint64)
        self.rids = rids.numpy()[:int(self.roff[-1])].astype(np.uint32)
        assert int(self.roff[-1]) <= max_ids

    def dist_counts_device(self, qoff, qids, n_qry, q_begin, q_end, counts, stream=None):
        """kssd_gpu_dist_counts_device: row q of [q_begin, q_end) at (q - q_begin) * n_ref"""
        qo = qoff.numpy()[:n_qry + 1].astype(np.uint64)
        qi = qids.numpy()[:int(qo[-1])].astype(np.uint32)
        full = ko.shared_counts(self.roff, self.rids, qo, qi)
        counts.view(-1)[:(q_end - q_begin) * full.shape[1]] = torch.from_numpy(full[q_begin:q_end].astype(np.int32).reshape(-1))

    def transpose_metrics_device(self, qoff, n_qry, q_begin, q_end, counts, out_pitch, shared_t, *planes, stream=None):
        """kssd_gpu_transpose_metrics_device: element (indexed sketch r, query q) at r * out_pitch + (q - q_begin)"""
        n_ref, rows = len(self.roff) - 1, q_end - q_begin
        c = counts.view(-1)[:rows * n_ref].view(rows, n_ref)
        flat = shared_t.view(-1)
        for r in range(n_ref):
            flat[r * out_pitch:r * out_pitch + rows] = c[:, r]

    def dist_device_transposed(self, qoff, qids, n_qry, q_begin, q_end, work, out_pitch, shared_t, *planes, stream=None):
        """kssd_gpu_dist_device_transposed: element (indexed sketch r, query q) at r * out_pitch + (q - q_begin)"""
        qo = qoff.numpy()[:n_qry + 1].astype(np.uint64)
        qi = qids.numpy()[:int(qo[-1])].astype(np.uint32)
        full = ko.shared_counts(self.roff, self.rids, qo, qi)          # [n_qry] x [n_ref]
        n_ref, rows = full.shape[1], q_end - q_begin
        assert out_pitch >= rows and work.numel() >= rows * n_ref
        flat = shared_t.view(-1)
        for r in range(n_ref):
            flat[r * out_pitch:r * out_pitch + rows] = torch.from_numpy(full[q_begin:q_end, r].astype(np.int32))

    def dist_device(self, qoff, qids, n_qry, q_begin, q_end, shared, *planes, stream=None):
        qo = qoff.numpy()[:n_qry + 1].astype(np.uint64)
        qi = qids.numpy()[:int(qo[-1])].astype(np.uint32)
        full = ko.shared_counts(self.roff, self.rids, qo, qi)
        shared.view(-1)[:(q_end - q_begin) * full.shape[1]] = torch.from_numpy(full[q_begin:q_end].astype(np.int32).reshape(-1))


def _search_worker(rank, world, port, G, cap, Qn, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from public_kssd_amd.shard import ShardedSearch
        off, ids = make_rank_data(rank, G, cap)
        qoff, qids = make_rank_data(1000 + rank, Qn, cap)
        out = {}
        for part in ("query", "own"):
            s = ShardedSearch(world, rank, G, cap, torch.device("cpu"), OracleEngine(), partition=part)
            shared = torch.zeros(s.cells(), dtype=torch.int32)
            s.step(torch.from_numpy(off), torch.from_numpy(ids), shared, None, cap)
            out[part] = (shared.numpy().copy(), s.block())
            assert s.block() == ((rank * G, (rank + 1) * G), (0, world * G), False)   # either partition leaves the rank's rows, row-major
        # a query set of its own (Q != R): only the north_star partition can do it
        s = ShardedSearch(world, rank, G, cap, torch.device("cpu"), OracleEngine(), partition="query")
        shared = torch.zeros(s.cells(Qn), dtype=torch.int32)
        s.step(torch.from_numpy(off), torch.from_numpy(ids), shared, None, cap,
               q=(torch.from_numpy(qoff), torch.from_numpy(qids), Qn))
        out["search"] = (shared.numpy().copy(), s.block(Qn))
        try:
            ShardedSearch(world, rank, G, cap, torch.device("cpu"), OracleEngine(), partition="own").step(
                torch.from_numpy(off), torch.from_numpy(ids), shared, None, cap, q=(torch.from_numpy(qoff), torch.from_numpy(qids), Qn))
            out["refused"] = False
        except ValueError:
            out["refused"] = True
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def _csr_of(seed0, world, G, cap):
    offs, idl = [0], []
    for r in range(world):
        off, ids = make_rank_data(seed0 + r, G, cap)
        for g in range(G):
            idl.append(ids[off[g]:off[g + 1]])
            offs.append(offs[-1] + int(off[g + 1] - off[g]))
    return np.array(offs, dtype=np.uint64), np.concatenate(idl).astype(np.uint32)


@pytest.mark.parametrize("world", [2, 3])
def test_blocks_of_both_partitions_assemble_into_the_oracle_matrix(world):
    """every rank computes its block (north_star partition: own query rows x gathered full index; own-index partition:
    all gathered rows x own index, written transposed); the blocks put together must be the oracle's full matrix -- for
    all-pairs in both partitions, and for a query set of its own (Q != R) in the north_star partition"""
    from public_kssd_amd.shard import assemble
    G, cap, Qn = 13, 700, 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_search_worker, args=(r, world, port, G, cap, Qn, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    roff, rids = _csr_of(0, world, G, cap)
    want = ko.shared_counts(roff, rids, roff, rids)
    for part in ("query", "own"):
        full = assemble(world, G, [res[r][part] for r in range(world)])
        assert np.array_equal(full.astype(np.uint32), want), part
        # the layout `kssd dist --allpairs --gpus N` relies on (csrc/kssd_resident.inc): rank r's flat output IS rows
        # [r*G, (r+1)*G) of the row-major N x N matrix -- the ranks' outputs one behind the other are sharedk_ct.dat
        cat = np.concatenate([res[r][part][0] for r in range(world)]).astype(np.uint32)
        assert np.array_equal(cat, want.reshape(-1)), part
    qoff, qids = _csr_of(1000, world, Qn, cap)
    want_q = ko.shared_counts(roff, rids, qoff, qids)
    full = assemble(world, G, [res[r]["search"] for r in range(world)], Q=Qn)
    assert full.shape == (world * Qn, world * G) and np.array_equal(full.astype(np.uint32), want_q)
    assert all(res[r]["refused"] for r in range(world))
