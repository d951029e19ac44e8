#!/usr/bin/env python3
"""bench.py -- kssd hot path on MI355X: genomes sketched/s (+ pairwise distances/s) at L3K10.

One step = one pass of the hot path over one batch that is already resident in HBM:
    sketch (scan + per-genome dedup)  ->  [N>1: RCCL all-gather of the packed sketches]
    ->  inverted index of all reference sketches  ->  all-pairs rows of this rank's query block
        (shared counts + Jaccard / MashD / containment / AafD, 36 B per pair)
Workload = BASELINE.json configs[1]: 1 000 synthetic 5 Mb bacterial genomes per GPU (50 clades x 20 members,
0.5-5 % substitutions, 1e-4 N), L3K10 shuffle, all-pairs.  Weak scaling: every rank sketches its own 1 000
genomes and computes the block of the all-pairs matrix between ITS genomes and ALL N x 1 000 genomes.  It puts
its own genomes on the indexed side (all gathered sketches are the query rows): the index build is the part
that would otherwise be repeated on every rank, and every metric of the path is symmetric in (query, reference),
so the R x G block a rank writes is the transpose of its G x R query block.

By default the steps run back to back on one stream (--inflight 1): the roofline figure of the scan is then the kernel's
own.  --inflight 3 pipelines the steps over three contexts and streams, so that the latency-bound kernels of one step
(exact stage, index build) run underneath the scan of the next one: +8 % genomes/s on one MI355X, at the price of a scan
that shares the machine (its launch takes a few per cent longer); DESIGN.md section 5 has both sets of numbers and the
other schedules that were measured.

    python bench.py --gpus 1 --steps 10 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import public_kssd_amd as K  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def log(*a):
    print(*a, file=sys.stderr, flush=True)


# ------------------------------------------------------------------------------------------------------
# synthetic batch, generated and packed on the device (setup, untimed)
# ------------------------------------------------------------------------------------------------------
def make_batch(n_genomes, length, n_clades, seed, dev, keep_codes=0, on_genome=None, keep_on_device=False):
    """returns packed int32[words+slack], mask int32[...], chunk_off uint64[n+1], kept [(codes u8, nmask bool)]
    on_genome(gi, codes u8 tensor, nmask bool tensor): called for every genome (device tensors, valid during the call);
    keep_on_device: `kept` holds device tensors instead of numpy arrays"""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    chunks = (length + K.CHUNK_BASES - 1) // K.CHUNK_BASES
    padded = chunks * K.CHUNK_BASES
    packed = torch.zeros(n_genomes * chunks * K.CHUNK_WORDS + 64, dtype=torch.int32, device=dev)
    mask = torch.zeros(n_genomes * chunks * K.CHUNK_MASKW + 64, dtype=torch.int32, device=dev)
    wsh = (30 - 2 * torch.arange(16, device=dev, dtype=torch.int64))
    msh = torch.arange(32, device=dev, dtype=torch.int64)
    per = (n_genomes + n_clades - 1) // n_clades
    kept = []
    gi = 0
    for c in range(n_clades):
        anc = torch.randint(0, 4, (length,), generator=g, device=dev, dtype=torch.uint8)
        for m in range(per):
            if gi >= n_genomes:
                break
            rate = 0.005 + 0.045 * float(torch.rand((), generator=g, device=dev))
            mut = torch.rand(length, generator=g, device=dev) < rate
            add = torch.randint(1, 4, (length,), generator=g, device=dev, dtype=torch.uint8)
            codes = torch.where(mut, (anc + add) & 3, anc)
            nmask = torch.rand(length, generator=g, device=dev) < 1e-4
            valid = ~nmask
            codes_v = torch.where(valid, codes, torch.zeros_like(codes))
            cp = torch.zeros(padded, dtype=torch.int64, device=dev)
            cp[:length] = codes_v
            vp = torch.zeros(padded, dtype=torch.int64, device=dev)
            vp[:length] = valid
            w = (cp.view(-1, 16) << wsh).sum(1)
            mw = (vp.view(-1, 32) << msh).sum(1)
            packed[gi * chunks * K.CHUNK_WORDS:(gi + 1) * chunks * K.CHUNK_WORDS] = w.to(torch.int32)
            mask[gi * chunks * K.CHUNK_MASKW:(gi + 1) * chunks * K.CHUNK_MASKW] = mw.to(torch.int32)
            if gi < keep_codes:
                kept.append((codes, nmask) if keep_on_device else (codes.cpu().numpy(), nmask.cpu().numpy()))
            if on_genome is not None:
                on_genome(gi, codes, nmask)
            gi += 1
    chunk_off = np.arange(n_genomes + 1, dtype=np.uint64) * np.uint64(chunks)
    return packed, mask, chunk_off, kept


# ------------------------------------------------------------------------------------------------------
# CPU baseline (rank 0, N=1, bounded sample)
# ------------------------------------------------------------------------------------------------------
def cpu_baseline(shuf, kept, cores, gpu_sets):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import kssd_oracle as ko
    from synth import fasta_text
    texts = [fasta_text(c, b"g%d" % i, n_mask=m) for i, (c, m) in enumerate(kept)]
    nb = sum(len(c) for c, _ in kept)
    out = {}
    # the port (our C restatement of the reference algorithm), OpenMP over genomes like run_stageI
    t0 = time.time()
    off, ids = ko.sketch_texts(shuf.table, shuf.k, shuf.subk, shuf.drlevel, texts, threads=cores)
    t_port = time.time() - t0
    # parity of the bench inputs while we are here: the GPU sketches of the sample must equal the oracle's
    for g in range(len(texts)):
        want = np.sort(ids[int(off[g]):int(off[g + 1])])
        assert np.array_equal(gpu_sets[g], want), "bench sample genome %d: GPU sketch != oracle" % g
    port = {"value": len(texts) / t_port, "unit": "genomes/s", "cores": min(cores, len(texts)), "kind": "port",
            "sample": "%d of the bench genomes (%.0f Mbase) as 70-col FASTA text in memory, oracle/kssd_oracle.c "
                      "sketch_texts, OpenMP over genomes" % (len(texts), nb / 1e6),
            "mbase_per_s": nb / 1e6 / t_port}
    out["port"] = port
    # the real reference binary when the snapshot carries it
    if ko.have_ref() and shutil.which("zcat"):
        d = tempfile.mkdtemp(prefix="kssd_bench_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
        try:
            os.mkdir(os.path.join(d, "fa"))
            for i, t in enumerate(texts):
                with open(os.path.join(d, "fa", "g%04d.fasta" % i), "wb") as f:
                    f.write(t)
            shuf.write(os.path.join(d, "L3K10.shuf"))
            t0 = time.time()
            # the reference only goes parallel when there are more files than threads (command_dist.c:275)
            p_ref = max(1, min(cores, len(texts) - 1))
            ko.run_ref(["dist", "-p", p_ref, "-L", "L3K10.shuf", "-o", "sk", "fa"], cwd=d, timeout=900)
            t_ref = time.time() - t0
            sets = ko.sketch_sets_by_name(os.path.join(d, "sk"))
            for i in range(len(texts)):
                assert np.array_equal(sets["g%04d.fasta" % i], gpu_sets[i]), "reference binary sketch != GPU sketch"
            out["reference"] = {"value": len(texts) / t_ref, "unit": "genomes/s", "cores": p_ref, "kind": "reference",
                                "sample": "%d of the bench genomes (%.0f Mbase) as FASTA files in tmpfs, "
                                          "`oracle/_ref/kssd dist -p %d -L L3K10.shuf` wall time incl. process start "
                                          "and the 64 MiB .shuf load" % (len(texts), nb / 1e6, p_ref),
                                "mbase_per_s": nb / 1e6 / t_ref}
        finally:
            shutil.rmtree(d, ignore_errors=True)
    # distances: posting traversal + output_ctrl arithmetic of the port on the sample's all-pairs
    t0 = time.time()
    sh = ko.shared_counts(off, ids, off, ids, threads=cores)
    t_cnt = time.time() - t0
    out["dist_port"] = {"value": sh.size / t_cnt, "unit": "pairs/s", "cores": cores, "kind": "port",
                        "sample": "%dx%d all-pairs of the sample sketches, index build + posting traversal only "
                                  "(the reference adds a fixed ~7-25 s for its 2 GiB mco.index and ~2 us/pair of "
                                  "text formatting)" % (len(texts), len(texts))}
    return out


# ------------------------------------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--genomes", type=int, default=1000, help="genomes per GPU")
    ap.add_argument("--length", type=int, default=5_000_000)
    ap.add_argument("--clades", type=int, default=50)
    ap.add_argument("--cpu-sample", type=int, default=128, help="genomes of the CPU baseline sample (0 = skip)")
    ap.add_argument("--no-planes", action="store_true", help="shared counts only (4 B/pair instead of 36)")
    ap.add_argument("--spinup", type=int, default=40, help="untimed steps before the warmup steps (GPU clock ramp; 0 = none)")
    ap.add_argument("--inflight", type=int, default=int(os.environ.get("KSSD_BENCH_INFLIGHT", "1")),
                    help="batches in flight, each on its own HIP stream with its own context and outputs (1 = serial, the default; 3 = pipelined)")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d"
                         % (a.gpus, world, a.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU path to measure)")
    if os.environ.get("KSSD_BENCH_ONE_DEVICE"):  # development: every rank on cuda:0 (checks the N > 1 flow on a 1-GPU box)
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("KSSD_BENCH_BACKEND", "nccl")  # nccl = RCCL over xGMI; gloo only for the 1-GPU check
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    G, L = a.genomes, a.length
    NF = max(1, a.inflight)
    shuf = K.Shuf.generate(10, 6, 3, seed=20260101)
    t0 = time.time()
    n_keep = a.cpu_sample if (rank == 0 and world == 1) else 0
    packed, mask, chunk_off, kept = make_batch(G, L, a.clades, 20260101 + 7919 * rank, dev, keep_codes=min(n_keep, G))
    torch.cuda.synchronize()
    if rank == 0:
        log("[bench] batch of %d x %.1f Mb packed on device in %.1f s" % (G, L / 1e6, time.time() - t0))

    # Optional software pipeline over steps (--inflight 3).  Every step is the whole hot path over one batch; NF batches
    # are in flight, each with its own context (workspaces, index), outputs and HIP stream.  The scan holds every CU's
    # LDS, so no other LDS user starts while it runs; events between the phases (kssd_gpu_sketch_phase) let the
    # kernels without LDS of the neighbouring steps (exact stage, index insert) run underneath it and the LDS users
    # (per-genome sort, all-pairs rows, posting allocation) side by side between two scans.
    # Nothing is skipped or reused between steps; --inflight 1 (the default) runs the same phases back to back on one
    # stream.
    exp_ids = int(G * L / 4096)
    cap = int(exp_ids * 1.25) + 4096                      # ids per rank (padded all-gather unit)
    R = G * world
    if world > 1:
        from public_kssd_amd.shard import SketchGather

    class Slot:
        pass
    slots = []
    for j in range(NF):
        sl = Slot()
        sl.ctx = K.GpuCtx(shuf, local)
        # outputs, all preallocated: nothing is allocated inside the timed region after the warmup
        sl.off_l = torch.zeros(G + 1, dtype=torch.int64, device=dev)
        sl.ids_l = torch.zeros(cap, dtype=torch.int32, device=dev)
        sl.shared = torch.zeros(G * R, dtype=torch.int32, device=dev)
        sl.planes = [None] * 4 if a.no_planes else [torch.zeros(G * R, dtype=torch.float64, device=dev) for _ in range(4)]
        sl.tstream = torch.cuda.Stream(device=dev) if NF > 1 else torch.cuda.current_stream()
        sl.stream = sl.tstream.cuda_stream
        sl.scanned, sl.sorted = torch.cuda.Event(), torch.cuda.Event()
        sl.gather = SketchGather(world, G, cap, dev) if world > 1 else None
        # upper bound of the ids the index has to hold (it sizes the hash table): the padded capacity until the first
        # status read-back has told the host how many ids this batch really has
        sl.idx_bound = cap
        slots.append(sl)

    def plan_prep(sl):
        sl.ctx.sketch_plan(packed, mask, chunk_off, sl.off_l, sl.ids_l, cap, K.SKETCH_FASTA, 1)
        sl.ctx.sketch_phase(K.PHASE_PREP, sl.stream)

    def index_build(sl):
        if world == 1:
            sl.q = (sl.off_l, sl.ids_l)
        else:
            # the one exchange step of the path: all-gather of every rank's packed sketches (RCCL over xGMI),
            # fixed-size padded units compacted on the device, so that no size has to visit the host
            with torch.cuda.stream(sl.tstream):
                sl.q = sl.gather(sl.off_l, sl.ids_l)
        # index this rank's own sketches only (constant work per rank), query with everybody's: the R x G block
        # [all genomes] x [this rank's genomes] = transpose of rows [rank*G, (rank+1)*G) of the global matrix
        sl.ctx.index_build_device(sl.off_l, sl.ids_l, G, sl.idx_bound, sl.stream)

    def rows(sl):
        sl.ctx.dist_device(sl.q[0], sl.q[1], R, 0, R, sl.shared, *sl.planes, stream=sl.stream)

    def run_steps(n_steps):
        """n_steps whole steps, pipelined over the NF slots; everything is enqueued, nothing synchronised"""
        if NF < 3:
            for n in range(n_steps):                      # back to back (NF = 2: two independent chains)
                sl = slots[n % NF]
                plan_prep(sl)
                for ph in (K.PHASE_SCAN, K.PHASE_EXACT, K.PHASE_FINISH):
                    sl.ctx.sketch_phase(ph, sl.stream)
                index_build(sl)
                rows(sl)
            return
        if n_steps > 0:
            plan_prep(slots[0])
        for n in range(n_steps + 2):                      # two more rounds drain the pipeline
            cur, prev, old = slots[n % NF], slots[(n - 1) % NF], slots[(n - 2) % NF]
            if n < n_steps:
                if n >= 2:
                    cur.tstream.wait_event(old.sorted)    # the gap's LDS users are through (the rows of step n-3
                cur.ctx.sketch_phase(K.PHASE_SCAN, cur.stream)   # precede this scan on its own stream)
                cur.scanned.record(cur.tstream)
            if 1 <= n <= n_steps:                         # step n-1: exact stage under scan n, sort after it
                prev.ctx.sketch_phase(K.PHASE_EXACT, prev.stream)
                if n < n_steps:
                    prev.tstream.wait_event(cur.scanned)
                prev.ctx.sketch_phase(K.PHASE_FINISH, prev.stream)
                prev.sorted.record(prev.tstream)
                index_build(prev)                         # runs under scan n+1
            if n + 1 < n_steps:                           # setup of step n+1 ahead of the rows that share its stream
                plan_prep(slots[(n + 1) % NF])
            if 2 <= n:                                    # step n-2: all-pairs rows after scan n
                if n < n_steps:
                    old.tstream.wait_event(cur.scanned)
                rows(old)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # first calls size the workspaces of every context; retry if a staging region was too small
    for sl in slots:
        for attempt in range(6):
            plan_prep(sl)
            for ph in (K.PHASE_SCAN, K.PHASE_EXACT, K.PHASE_FINISH):
                sl.ctx.sketch_phase(ph, sl.stream)
            index_build(sl)
            rows(sl)
            rc, total, bad = sl.ctx.sketch_status(sl.stream)
            if rc == 0:
                sl.idx_bound = min(cap, int(total) + 1024)
                break
            if rc != K.capi.ERR_OVERFLOW:
                raise SystemExit("sketch failed: rc=%d" % rc)
        else:
            raise SystemExit("sketch kept overflowing")
    sync()
    # setup, untimed like the sizing passes above: the clocks of an idle GPU need some tens of milliseconds of work to
    # settle (measured: the scan launch takes 0.54 ms in the first dozen steps after a pause and 0.52 ms from then on)
    run_steps(a.spinup)
    sync()
    run_steps(a.warmup)
    sync()
    for sl in slots:
        sl.ctx.kernel_time(0, reset=True)
        sl.ctx.kernel_time(1, reset=True)
    t0 = time.perf_counter()
    run_steps(a.steps)
    sync()
    dt = time.perf_counter() - t0
    scan_ms = dist_ms = 0.0
    scan_n = dist_n = 0
    used = slots
    for sl in used:
        rc, total, bad = sl.ctx.sketch_status(sl.stream)
        if rc != 0:
            raise SystemExit("sketch status rc=%d after the timed loop" % rc)
        ms, n = sl.ctx.kernel_time(0)
        scan_ms += ms * n
        scan_n += n
        ms, n = sl.ctx.kernel_time(1)
        dist_ms += ms * n
        dist_n += n
    scan_ms = scan_ms / scan_n if scan_n else 0.0       # average launch duration over every timed launch
    dist_ms = dist_ms / dist_n if dist_n else 0.0
    ctx, stream = slots[0].ctx, slots[0].stream
    off_l, ids_l, shared = slots[0].off_l, slots[0].ids_l, slots[0].shared
    n_stage1, n_bloom = ctx.scan_stats(stream)
    for sl in used[1:]:                                   # every slot worked on the same batch: same results
        assert torch.equal(sl.off_l, off_l) and torch.equal(sl.shared, shared), "slots disagree"
        assert torch.equal(sl.ids_l[:int(total)], ids_l[:int(total)]), "slots disagree"

    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    # outside the timed region: the scan with nothing else on the device (sketch call alone, one stream), for the
    # kernel's own roofline figure next to the one measured under the pipeline's contention
    scan_alone_ms = None
    if NF > 1 and world == 1:
        sl = slots[0]
        sl.ctx.kernel_time(0, reset=True)
        for _ in range(5):
            plan_prep(sl)
            for ph in (K.PHASE_SCAN, K.PHASE_EXACT, K.PHASE_FINISH):
                sl.ctx.sketch_phase(ph, sl.stream)
        torch.cuda.synchronize()
        scan_alone_ms, _ = sl.ctx.kernel_time(0)

    if rank == 0:
        # size-independent sanity on the full matrix of this rank
        sh = shared.view(R, G)  # [all genomes (query rows)] x [this rank's genomes]; world 1: the full G x G matrix
        szs = (off_l[1:] - off_l[:-1]).to(torch.int32)
        diag = sh[rank * G + torch.arange(G, device=dev), torch.arange(G, device=dev)]
        assert torch.equal(diag, szs), "diagonal of the all-pairs matrix must equal the sketch sizes"
        own = sh[rank * G:(rank + 1) * G, :]
        assert torch.equal(own, own.t()), "all-pairs shared-count matrix must be symmetric"
        n_bases = G * L
        scan_bytes = 0.375 * n_bases + 4.0 * total        # 2-bit base + 1-bit mask, 4 B per id
        achieved = scan_bytes / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0
        pairs = G * R
        dist_bytes = (4 if a.no_planes else 36) * pairs + 4.0 * (total + R / G * total)
        res = {
            "metric": "genomes sketched/s (whole hot path per step: sketch + index + all-pairs distances, L3K10)",
            "value": world * G * a.steps / dt,
            "unit": "genomes/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: %d synthetic %.1f Mb bacterial genomes per GPU (%d clades), "
                                   "L3K10 sketch + all-pairs" % (G, L / 1e6, a.clades),
                       "k": 10, "subk": 6, "drlevel": 3, "genomes_per_gpu": G, "genome_len": L,
                       "pairs_per_step": world * pairs, "batches_in_flight": NF, "parallelism": "genomes and matrix blocks sharded x%d (own genomes "
                       "indexed, all gathered sketches as query rows), all-gather of sketches" % world if world > 1
                       else "single GPU"},
            "pairs_per_s": world * pairs * a.steps / dt,
            "mbase_per_s": world * n_bases * a.steps / dt / 1e6,
            "ids_per_batch": int(total),
            "kernels": {"sketch_scan_ms": scan_ms, "dist_rows_ms": dist_ms, "launches_timed": [scan_n, dist_n],
                        "dist_rows_GBs": dist_bytes / (dist_ms * 1e-3) / 1e9 if dist_ms > 0 else None,
                        "scan_positions_past_stage1": n_stage1 / n_bases, "scan_positions_past_bloom": n_bloom / n_bases},
            "roofline": {"bound": "hbm", "kernel": "sketch_scan_kernel<6>", "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": None,
                         "algorithmic_bytes_per_launch": scan_bytes},
        }
        if scan_alone_ms:
            res["kernels"]["sketch_scan_ms_alone"] = scan_alone_ms
            res["roofline"]["frac_alone"] = scan_bytes / (scan_alone_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
        if a.cpu_sample and world == 1 and kept:
            ol = off_l.cpu().numpy()
            il = ids_l.cpu().numpy().view(np.uint32)
            gpu_sets = [il[int(ol[g]):int(ol[g + 1])] for g in range(len(kept))]
            cores = os.cpu_count() or 1
            cb = cpu_baseline(shuf, kept, cores, gpu_sets)
            res["cpu_baseline"] = cb.get("reference", cb["port"])
            res["cpu_baseline_port"] = cb["port"]
            res["cpu_baseline_dist"] = cb["dist_port"]
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc) and G == 1000 and L == 5_000_000:  # the PMC passes were collected on the default workload
            try:
                res["roofline"]["traffic"] = json.load(open(pmc)).get("sketch_scan_bytes_per_launch")
            except Exception:
                pass
        print(json.dumps(res), flush=True)
    for sl in slots:
        sl.ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
