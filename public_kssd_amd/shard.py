"""torch.distributed plumbing of the multi-GPU path (one process per GPU, RCCL over xGMI on the GPU box, gloo in
the CPU tests).  The path shards by genome (sketching: no communication) and by block of the distance matrix; its
ONE exchange step is an all-gather of every rank's packed sketches.

Two partitions of the distance matrix (reference: one owner per output row,
command_dist.c:774-785):

  "query"      the north_star partition, valid for any query set: every rank gathers ALL reference sketches, builds
               the full index, and computes the rows of ITS OWN query block against all references.  Rank r writes
               rows [r*Q, (r+1)*Q) x all columns.  Per-rank cost at N ranks: index over N x the ids, Q probes rows.
  "own"        all-pairs only (queries = references, every metric of the path is symmetric): every rank indexes
               only ITS OWN sketches, runs all gathered sketches as query rows and writes the block TRANSPOSED
               (kssd_gpu_dist_device_transposed: counts row-major by query, then a tiled transpose that computes the
               metrics) -- so what it leaves is what "query" leaves: rows [r*G, (r+1)*G) x all columns, row-major,
               one owner per output row.  The index build, the part "query" repeats on every rank, stays constant
               per rank; the price is N x the probes, most of which miss (a negative filter in front of the table).
               ("transpose" is the name rounds 2-4 used for it, when the block was left as [all] x [own].)

Everything is static-shape tensor code on the caller's device: no size ever visits the host, so a step stays free
of host synchronisation.  The compute calls go through an `engine` with the two device-level entry points of the
C ABI (public_kssd_amd.GpuCtx: index_build_device / dist_device); the CPU tests plug in an oracle-backed engine.
"""
import torch


class SketchGather:
    """All-gather fixed-size padded sketch units and compact them into one CSR on the device.

    every rank contributes   off_l  int64[G+1]  exclusive prefix of its G sketch sizes
                             ids_l  int32[cap]  its ids, only the first off_l[G] are meaningful
    and receives             roff   int64[world*G+1], rids int32[world*cap] (first roff[-1] meaningful):
    genome r*G+g of the global numbering is genome g of rank r.
    """

    def __init__(self, world, G, cap, device, engine=None):
        """engine: an object with concat_units_device (public_kssd_amd.GpuCtx) does the unpacking with one copy kernel;
        without it (CPU tests) the same CSR comes out of tensor ops"""
        self.world, self.G, self.cap = world, G, cap
        self.engine = engine if hasattr(engine, "concat_units_device") else None
        self.off_all = torch.zeros(world * (G + 1), dtype=torch.int64, device=device)
        self.ids_all = torch.zeros(world * cap, dtype=torch.int32, device=device)
        self.roff = torch.zeros(world * G + 1, dtype=torch.int64, device=device)
        self.rids = torch.zeros(world * cap, dtype=torch.int32, device=device)
        self._j = torch.arange(world * cap, device=device, dtype=torch.int64)

    def __call__(self, off_l, ids_l, group=None, stream=None):
        import torch.distributed as dist
        dist.all_gather_into_tensor(self.off_all, off_l, group=group)
        dist.all_gather_into_tensor(self.ids_all, ids_l, group=group)
        if self.engine is not None:
            self.engine.concat_units_device(self.off_all, self.ids_all, self.world, self.G, self.cap, self.roff, self.rids, stream)
            return self.roff, self.rids
        return self.compact()

    def compact(self):
        w, G, cap = self.world, self.G, self.cap
        o = self.off_all.view(w, G + 1)
        sizes = (o[:, 1:] - o[:, :-1]).reshape(-1)
        self.roff[0] = 0
        self.roff[1:] = torch.cumsum(sizes, 0)
        tot = o[:, G]                                   # ids held by each rank
        ends = torch.cumsum(tot, 0)
        starts = ends - tot
        rk = torch.searchsorted(ends, self._j, right=True).clamp_(max=w - 1)
        src = (self._j - starts[rk] + rk * cap).clamp_(min=0, max=w * cap - 1)
        torch.index_select(self.ids_all, 0, src, out=self.rids)
        return self.roff, self.rids


def query_block(rank, G):
    """rows of the global matrix rank `rank` owns when every rank holds G queries"""
    return rank * G, (rank + 1) * G


class ShardedSearch:
    """One rank's share of the distance matrix between `world` x G reference sketches (G contributed by every rank)
    and the queries, in either partition (see the module docstring).

        s = ShardedSearch(world, rank, G, cap, device, engine, partition="query")
        s.step(off_l, ids_l, shared, planes, max_ids)                      # all-pairs: the rank's queries = its refs
        s.step(off_l, ids_l, shared, planes, max_ids, q=(qoff, qids, Q))   # "query" only: any local query block
        rows, cols, transposed = s.block(Q)     # where `shared` sits in the global matrix

    `shared` (and each plane) is a flat tensor the caller preallocates with s.cells(Q) elements.
    gather: the exchange; the default is SketchGather (torch.distributed).  bench.py --emulate-world plugs in one that
    delivers the other ranks' units by device-to-device copies.
    check_index: read the index build's status back and repeat an overflowed build (synchronises; GpuCtx's default).
    A timed loop sets it False and polls engine.index_status() itself.
    """

    def __init__(self, world, rank, G, cap, device, engine, partition="query", gather=None, check_index=True):
        if partition == "transpose":
            partition = "own"
        if partition not in ("query", "own"):
            raise ValueError("partition must be 'query' or 'own'")
        self.world, self.rank, self.G, self.cap = world, rank, G, cap
        self.engine, self.partition, self.check_index = engine, partition, check_index
        self.gather = gather if gather is not None else (SketchGather(world, G, cap, device, engine) if world > 1 else None)
        self._filter = world > 1 and partition == "own" and hasattr(engine, "index_set_filter")
        if self._filter:
            # (world - 1) / world of the query rows are other ranks' sketches and share next to nothing with the own index:
            # a negative filter in front of the table for them, the own block of rows exempt
            engine.index_set_filter(True, rank * G, (rank + 1) * G)
        # "own" on GPUs: the exchange runs on a stream of its own, under the index build and the rank's OWN block of
        # rows -- neither needs anybody else's sketches; only the foreign rows wait for it
        self._overlap = world > 1 and partition == "own" and torch.device(device).type == "cuda"
        if self._overlap:
            self._cstream = torch.cuda.Stream(device=device)
            self._ev_ready, self._ev_gathered = torch.cuda.Event(), torch.cuda.Event()
        # the counts of the rows in flight, row-major by query, before they are turned around (own-index partition)
        self._work = torch.zeros(world * G * G, dtype=torch.int32, device=device) if partition == "own" and world > 1 else None

    def cells(self, Q=None):
        Q = self.G if Q is None else Q
        return Q * self.G * self.world

    def block(self, Q=None):
        """(row range, column range, transposed) of this rank's output in the global [world*Q] x [world*G] matrix: in either
        partition the rows of the rank's own block, row-major (transposed is always False since round 5: the own-index
        partition turns its block around on the device)"""
        Q = self.G if Q is None else Q
        return (self.rank * Q, (self.rank + 1) * Q), (0, self.world * self.G), False

    def _build(self, roff, rids, n_ref, max_ids, stream):
        if self.check_index:
            self.engine.index_build_device(roff, rids, n_ref, max_ids, stream)
        else:
            self.engine.index_build_device(roff, rids, n_ref, max_ids, stream, check=False)

    def index(self, off_l, ids_l, max_ids, stream=None, tstream=None, group=None):
        """the exchange (N > 1) and the index build of one step; nothing is synchronised unless check_index is set.
        max_ids: upper bound of the ids one rank holds (sizes the index).  tstream: the torch stream object that
        wraps `stream` (the collective is issued under it)."""
        w, G = self.world, self.G
        if stream is not None and tstream is not None and int(tstream.cuda_stream) != int(stream):
            raise ValueError("tstream must wrap `stream`: the engine calls and the collective have to be ordered on one stream")
        if self._overlap:
            if stream is not None and tstream is None and int(stream) != int(torch.cuda.current_stream().cuda_stream):
                raise ValueError("overlap mode: pass tstream= together with stream= (the exchange is ordered behind the rank's "
                                 "sketches by an event recorded on that stream)")
            cur = tstream if tstream is not None else torch.cuda.current_stream()
            self._ev_ready.record(cur)                 # the rank's sketches are complete
            self._cstream.wait_event(self._ev_ready)
            with torch.cuda.stream(self._cstream):
                roff, rids = self.gather(off_l, ids_l, group=group, stream=self._cstream.cuda_stream)
                self._ev_gathered.record(self._cstream)
            self._gathered = (roff, rids)
            self._cur = cur
            self._build(off_l, ids_l, G, max_ids, stream)
            return roff, rids
        if w == 1:
            roff, rids = off_l, ids_l
        elif tstream is not None:
            with torch.cuda.stream(tstream):
                roff, rids = self.gather(off_l, ids_l, group=group, stream=stream)
        else:
            roff, rids = self.gather(off_l, ids_l, group=group, stream=stream)
        self._gathered = (roff, rids)
        if self.partition == "query":   # full index on every rank
            self._build(roff, rids, w * G, max_ids * w, stream)
        else:                           # own index only
            self._build(off_l, ids_l, G, max_ids, stream)
        return roff, rids

    def rows(self, off_l, ids_l, shared, planes, q=None, stream=None):
        """the rank's block of the matrix against the index of the last index() call"""
        w, G, r = self.world, self.G, self.rank
        if q is not None and self.partition != "query":
            raise ValueError("a separate query set needs the 'query' partition (the own-index trick is all-pairs only)")
        pl = list(planes) if planes is not None else [None] * 4
        if self.partition == "query" or w == 1:   # own query block as rows (one rank: the two partitions are the same calls)
            qoff, qids, Q = (off_l, ids_l, G) if q is None else q
            self.engine.dist_device(qoff, qids, Q, 0, Q, shared, *pl, stream=stream)
            return
        # everybody's sketches as query rows against the OWN index: counts row-major by query into the work array (the rank's own
        # rows from its own sketches while the exchange is still under way, the others' behind it -- walked flat behind the negative
        # filter), then ONE launch that turns all of them around: query g of the global numbering is column g of the rank's G rows
        # (row g of the work array = query g of the global numbering, G counters wide)
        cnt = lambda qoff, qids, n, a, b, row0: self.engine.dist_counts_device(qoff, qids, n, a, b, self._work[row0 * G:], stream=stream)
        if self._overlap:
            if self._filter:
                self.engine.index_set_filter(True, 0, G)                     # (the own block: rows 0 .. G of this call)
            cnt(off_l, ids_l, G, 0, G, r * G)
            if self._filter:
                self.engine.index_set_filter(True, r * G, (r + 1) * G)
            self._cur.wait_event(self._ev_gathered)
            roff, rids = self._gathered
            if r > 0:
                cnt(roff, rids, w * G, 0, r * G, 0)
            if r + 1 < w:
                cnt(roff, rids, w * G, (r + 1) * G, w * G, (r + 1) * G)
        else:
            roff, rids = self._gathered
            cnt(roff, rids, w * G, 0, w * G, 0)
        self.engine.transpose_metrics_device(roff, w * G, 0, w * G, self._work, w * G, shared, *pl, stream=stream)

    def step(self, off_l, ids_l, shared, planes, max_ids, q=None, stream=None, tstream=None, group=None):
        """index() + rows(): the exchange, the index build and the rows of one step; nothing is synchronised"""
        if q is not None and self.partition != "query":
            raise ValueError("a separate query set needs the 'query' partition (the own-index trick is all-pairs only)")
        out = self.index(off_l, ids_l, max_ids, stream=stream, tstream=tstream, group=group)
        self.rows(off_l, ids_l, shared, planes, q=q, stream=stream)
        return out


def assemble(world, G, blocks, Q=None):
    """global [world*Q] x [world*G] matrix from the ranks' outputs: blocks[rank] = (flat array, (rows, cols, transposed))
    as ShardedSearch.block describes them.  numpy in, numpy out (used by the tests and by single-host gathers)."""
    import numpy as np
    Q = G if Q is None else Q
    first = np.asarray(blocks[0][0])
    full = np.zeros((world * Q, world * G), dtype=first.dtype)
    for rank in range(world):
        flat, ((r0, r1), (c0, c1), transposed) = blocks[rank]
        a = np.asarray(flat)
        if transposed:
            full[r0:r1, c0:c1] = a.reshape(c1 - c0, r1 - r0).T
        else:
            full[r0:r1, c0:c1] = a.reshape(r1 - r0, c1 - c0)
    return full
