#!/usr/bin/env python3
"""bench.py -- kssd hot path on MI355X: genomes sketched/s (+ pairwise distances/s) at L3K10.

One step = one pass of the hot path over one batch that is already resident in HBM:
    sketch (scan + per-genome dedup)  ->  [N>1: RCCL all-gather of the packed sketches]
    ->  inverted index of all reference sketches  ->  all-pairs rows of this rank's query block
        (shared counts + Jaccard / MashD / containment / AafD, 36 B per pair)
Workload = BASELINE.json configs[1]: 1 000 synthetic 5 Mb bacterial genomes per GPU (50 clades x 20 members,
0.5-5 % substitutions, 1e-4 N), L3K10 shuffle, all-pairs.  Weak scaling: every rank sketches its own 1 000
genomes and computes the rows of ITS genomes against ALL N x 1 000 genomes (query-block sharding of the matrix, row-major).
How a rank computes them: --partition own (the headline) indexes only its own sketches, runs all gathered sketches as query
rows and writes the block transposed (queries = references, every metric symmetric: the index build stays constant per
rank); --partition query builds the full index of all gathered sketches on every rank (what a search with Q != R needs).

The steps run back to back on one stream.  Every --kernel-timing-th launch of the scan / rows kernels inside the timed region
(at least ten of them) carries the events of the roofline figures; their minimum and maximum stand beside the mean.
--emulate-world N: one GPU plays one rank of N (everything of the rank's step except xGMI), both partitions.

    python bench.py --gpus 1 --steps 10 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import public_kssd_amd as K  # noqa: E402
from benchlib.launch import host_cores, log, self_launch  # noqa: E402
from benchlib.workloads import READ_LEN, FastaTextSink, make_batch, make_long_records, make_reads_batch, mask_summary, summary_clear_lanes  # noqa: E402
from benchlib.multi import EmulatedGather, run_exchange_c  # noqa: E402

from benchlib.legs import HBM_PEAK_GBS, KSSD_BIN, _run_ours, cpu_baseline, run_fastq, run_mammal, tokeniser_leg  # noqa: E402,F401
# (the synthetic workloads -- make_batch, make_reads_batch, make_long_records --, the launcher plumbing and the legs beside the
# headline -- CPU baseline + end-to-end commands, the tokeniser's leg, the fastq and mammal workloads -- live in benchlib/)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--genomes", type=int, default=None, help="genomes per GPU (allpairs: 1000; mammal: as many 3 Gb records as asked, default 8)")
    ap.add_argument("--length", type=int, default=None, help="bases per genome (allpairs: 5 000 000; mammal: 3 000 000 000)")
    ap.add_argument("--clades", type=int, default=50)
    ap.add_argument("--cpu-sample", type=int, default=128, help="genomes of the CPU baseline sample (0 = skip)")
    ap.add_argument("--no-planes", action="store_true", help="shared counts only (4 B/pair instead of 36)")
    ap.add_argument("--no-tok-leg", action="store_true",
                    help="allpairs, one GPU: skip roofline_tok (the device tokeniser on the batch's genomes as resident FASTA text: what `kssd dist` "
                         "runs on the device in front of the scan)")
    ap.add_argument("--no-mask-summary", action="store_true",
                    help="A/B: the scan streams the whole validity mask (rounds 1-5) instead of reading the batch's summary words "
                         "(one 64-bit word per chunk, written once when the batch is made resident) and the mask words of the lanes they name")
    ap.add_argument("--spinup", type=int, default=40, help="untimed steps before the warmup steps (GPU clock ramp; 0 = none)")
    ap.add_argument("--workload", choices=["allpairs", "fastq", "mammal"], default="allpairs",
                    help="allpairs = BASELINE configs[1] (the metric's config; configs[2] with --genomes 10000 --clades 500); "
                         "fastq = configs[3]: reads -> read-set sketch -> containment against the reference sketches; "
                         "mammal = configs[4]: 3 Gb records at -k 10 -s 7 -l 5, sketch throughput")
    ap.add_argument("--partition", choices=["both", "own", "query", "transpose"], default="both",
                    help="N > 1, how a rank computes the rows of the matrix it owns (its query block x all genomes, row-major, in either "
                         "case): own = it indexes ITS OWN sketches, runs all gathered sketches as query rows and writes the block "
                         "transposed (all-pairs: queries = references, every metric symmetric; the index build stays constant per rank) -- "
                         "the headline; query = the full index of all gathered sketches on every rank, own query block as rows (what a "
                         "search with Q != R needs); both (default) = own as the headline value, query measured beside it")
    ap.add_argument("--emulate-world", type=int, default=0, metavar="N",
                    help="ONE GPU plays ONE rank (--rank) of N at the per-GPU size: its own batch sketched every step, the other ranks' "
                         "units (sketched once, untimed) delivered by device-to-device copies of exactly the bytes the all-gather "
                         "delivers, then index + rows in both partitions.  Everything of a rank's step except xGMI; the line is "
                         "labelled an emulation and is not a --gpus N result")
    ap.add_argument("--rank", type=int, default=0, help="--emulate-world: which rank this GPU plays")
    ap.add_argument("--e2e-files", type=int, default=1024, help="files of the end-to-end / reference leg (the CPU sample under several names)")
    ap.add_argument("--e2e-search", type=int, default=1024, help="sketches of the end-to-end search leg (all-pairs among the first N files)")
    ap.add_argument("--e2e-sketch-large", type=int, default=8192, help="names of the stage-I leg where process start is amortised (hard links onto the sample's files; 0 = skip)")
    ap.add_argument("--e2e-search-large", type=int, default=4096, help="sketches of the second search leg, where start-up no longer dominates (0 = skip)")
    ap.add_argument("--reads", type=int, default=100_000_000, help="fastq workload: reads of 150 bp")
    ap.add_argument("--parity-reads", type=int, default=10_000_000, help="fastq workload: reads of the oracle slice (0 = skip)")
    ap.add_argument("--exchange", choices=["torch", "c"], default=os.environ.get("KSSD_BENCH_EXCHANGE", "torch"),
                    help="N > 1: torch = one process per GPU, torch.distributed (RCCL) all-gather (the default; under torch.distributed.run "
                         "rank 0 falls back to `c` when the process group cannot be set up); c = ONE process drives the N devices, the "
                         "exchange is kssd_gpu_allgather_sketches of the C ABI (RCCL inside libkssd_gpu.so, no torch.distributed)")
    ap.add_argument("--kernel-timing", type=int, default=int(os.environ.get("KSSD_BENCH_KERNEL_TIMING", "4")),
                    help="allpairs: every N-th launch of the scan / rows kernels inside the timed region carries the HIP events the roofline "
                         "figures come from (kssd_gpu_set_kernel_timing; 1 = every launch), lowered until at least ten launches of the timed "
                         "region carry them.  A bracketed dispatch does not overlap its neighbours in the stream, which costs the step "
                         "several microseconds per bracket (profiles/r04E_gaps.txt)")
    a = ap.parse_args()
    if a.partition == "transpose":
        a.partition = "own"
    if a.genomes is None:
        a.genomes = 8 if a.workload == "mammal" else 1000
    if a.length is None:
        a.length = 3_000_000_000 if a.workload == "mammal" else 5_000_000
    if a.emulate_world and (a.emulate_world < 2 or a.gpus != 1 or a.workload != "allpairs" or not 0 <= a.rank < a.emulate_world):
        raise SystemExit("--emulate-world N: N >= 2, --gpus 1, --workload allpairs, 0 <= --rank < N")

    if a.exchange == "c" and a.workload == "allpairs" and "WORLD_SIZE" not in os.environ and not a.emulate_world:
        if not torch.cuda.is_available() or torch.cuda.device_count() < a.gpus:
            raise SystemExit("--exchange c --gpus %d: %d device(s) visible" % (a.gpus, torch.cuda.device_count() if torch.cuda.is_available() else 0))
        return run_exchange_c(a, K.Shuf.generate(10, 6, 3, seed=20260101), a.gpus)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(a, __file__)                             # nothing has touched the GPU yet
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
    backend = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        want = os.environ.get("KSSD_BENCH_BACKEND", "nccl")  # nccl = RCCL over xGMI; gloo only for the 1-GPU check
        try:
            import datetime
            if want == "nccl":
                dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(seconds=300))
            else:
                dist.init_process_group(want, timeout=datetime.timedelta(seconds=300))
            probe = torch.ones(1, device=dev)
            dist.all_reduce(probe)                        # the first collective sets the communicator up: fail here, not in the timed loop
            torch.cuda.synchronize()
            if int(probe.item()) != world:
                raise RuntimeError("all_reduce of ones over %d ranks gave %s" % (world, probe.item()))
        except Exception as e:                            # the second way for an N > 1 run to succeed: rank 0 drives all devices itself
            if a.workload != "allpairs" or os.environ.get("KSSD_BENCH_ONE_DEVICE"):
                raise
            log("[bench] rank %d: torch.distributed (%s) is not usable: %s" % (rank, want, str(e)[:300]))
            try:
                dist.destroy_process_group()
            except Exception:
                pass
            if rank != 0:
                return 0
            log("[bench] rank 0 runs the step on all %d devices itself (--exchange c: kssd_gpu_allgather_sketches)" % world)
            return run_exchange_c(a, K.Shuf.generate(10, 6, 3, seed=20260101), world, reason=str(e)[:300])
        backend = dist.get_backend()                      # what the line reports: the transport that really ran

    G, L = a.genomes, a.length
    if a.workload == "mammal":
        if world != 1:
            # the path shards by genome with no exchange: N ranks = N independent replicas of this run
            log("[bench] --workload mammal on %d ranks: independent replicas, no collective" % world)
        return run_mammal(a, dev, world, rank)
    shuf = K.Shuf.generate(10, 6, 3, seed=20260101)
    if a.workload == "fastq":
        if world != 1:
            raise SystemExit("--workload fastq is a single-GPU run (one read set = one sketch)")
        return run_fastq(a, shuf, dev)
    # W ranks of G genomes each; this process is rank R_ of them.  --emulate-world: W is the emulated world, one process, no collective
    emu = a.emulate_world > 0
    W, R_ = (a.emulate_world, a.rank) if emu else (world, rank)
    t0 = time.time()
    n_keep = a.cpu_sample if (rank == 0 and W == 1) else 0
    tok_leg = rank == 0 and W == 1 and not a.no_tok_leg
    sink = FastaTextSink(G, L, dev) if tok_leg else None
    packed, mask, chunk_off, kept = make_batch(G, L, a.clades, 20260101 + 7919 * R_, dev, keep_codes=min(n_keep, G), on_genome=sink)
    torch.cuda.synchronize()
    if rank == 0:
        log("[bench] batch of %d x %.1f Mb packed on device in %.1f s" % (G, L / 1e6, time.time() - t0))

    exp_ids = int(G * L / 4096)
    cap = int(exp_ids * 1.25) + 4096                      # ids per rank (padded all-gather unit)
    R = G * W
    from public_kssd_amd.shard import ShardedSearch

    emu_units, emu_unit = {}, cap
    if emu:
        # the other ranks' sketches: their batches (the seeds a --gpus N run gives them) sketched once, untimed, one at a time
        t0 = time.time()
        ctx0 = K.GpuCtx(shuf, local)
        most = 0
        for j in range(W):
            if j == R_:
                pj, mj, cj = packed, mask, chunk_off
            else:
                pj, mj, cj, _ = make_batch(G, L, a.clades, 20260101 + 7919 * j, dev)
            off_j = torch.zeros(G + 1, dtype=torch.int64, device=dev)
            ids_j = torch.zeros(cap, dtype=torch.int32, device=dev)
            for attempt in range(6):
                ctx0.sketch_device(pj, mj, cj, off_j, ids_j, cap)
                rc, total, bad = ctx0.sketch_status()
                if rc == 0:
                    break
                if rc != K.capi.ERR_OVERFLOW:
                    raise SystemExit("sketch of emulated rank %d failed: rc=%d" % (j, rc))
            else:
                raise SystemExit("sketch kept overflowing")
            most = max(most, int(total))
            if j != R_:
                emu_units[j] = (off_j, ids_j)
                del pj, mj
                torch.cuda.empty_cache()
        ctx0.close()
        emu_unit = min(cap, (most + 4096 + 1023) // 1024 * 1024)   # what the fullest rank holds + a margin: the unit of a --gpus N run
        log("[bench] --emulate-world %d, rank %d: the %d other ranks' batches sketched once in %.1f s (unit %d ids)"
            % (W, R_, W - 1, time.time() - t0, emu_unit))

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    spinup_done = [0]   # untimed spin-up steps the last measurement made (reported in the line)

    def measure(partition):
        """the whole measurement (sizing passes, spin-up, warm-up, the timed steps, the size-independent checks of the result)
        in one partition of the matrix; returns the numbers and rank 0's tensors for the line"""
        ctx = K.GpuCtx(shuf, local)
        # outputs, all preallocated: nothing is allocated inside the timed region after the warmup
        off_l = torch.zeros(G + 1, dtype=torch.int64, device=dev)
        ids_l = torch.zeros(cap, dtype=torch.int32, device=dev)
        shared = torch.zeros(G * R, dtype=torch.int32, device=dev)
        planes = [None] * 4 if a.no_planes else [torch.zeros(G * R, dtype=torch.float64, device=dev) for _ in range(4)]
        tstream = torch.cuda.current_stream()
        stream = tstream.cuda_stream
        st = {"unit": emu_unit if emu else cap, "idx_bound": cap, "search": None}
        summ = None if a.no_mask_summary else mask_summary(ctx, mask, int(chunk_off[-1]), dev)   # part of the resident batch, like the mask itself

        def new_search():
            gather = EmulatedGather(W, R_, G, st["unit"], dev, ctx, emu_units) if emu else None
            st["search"] = ShardedSearch(W, R_, G, st["unit"], dev, ctx, partition=partition, gather=gather, check_index=False)
        new_search()

        def sketch():
            ctx.sketch_plan(packed, mask, chunk_off, off_l, ids_l, cap, K.SKETCH_FASTA, 1, d_summary=summ)
            for ph in (K.PHASE_PREP, K.PHASE_SCAN, K.PHASE_EXACT, K.PHASE_FINISH):
                ctx.sketch_phase(ph, stream)

        def index_build():
            # the one exchange step of the path (N > 1): all-gather of every rank's packed sketches (RCCL over xGMI), then the
            # index in the chosen partition (public_kssd_amd/shard.py).  The unit a rank contributes is its first `unit` ids.
            st["search"].index(off_l, ids_l[:st["unit"]], st["idx_bound"], stream=stream, tstream=tstream if W > 1 else None)

        def rows():
            st["search"].rows(off_l, ids_l[:st["unit"]], shared, planes, stream=stream)

        def run_steps(n_steps):
            """n_steps whole steps back to back on one stream; everything is enqueued, nothing synchronised"""
            for _ in range(n_steps):
                sketch()
                index_build()
                rows()

        # first calls size the workspaces of the context; retry if a staging region was too small
        for attempt in range(6):
            run_steps(1)
            rc, total, bad = ctx.sketch_status(stream)
            irc = ctx.index_status(stream)        # (a capped index build that overflowed: the next one counts first)
            if world > 1:
                # a step holds a collective: the ranks repeat it together or not at all (a rank whose staging regions were too small
                # while its neighbours' were not would otherwise meet their barrier with its all-gather)
                flags = torch.tensor([0 if (rc == 0 and irc == 0) else 1,
                                      0 if (rc in (0, K.capi.ERR_OVERFLOW) and irc in (0, K.capi.ERR_OVERFLOW)) else 1], dtype=torch.int32, device=dev)
                dist.all_reduce(flags, op=dist.ReduceOp.MAX)
                again, fatal = int(flags[0].item()), int(flags[1].item())
                if fatal:
                    raise SystemExit("sketch / index failed on some rank: this rank's rc=%d, %d" % (rc, irc))
                if again and rc == 0 and irc == 0:
                    continue
            if rc == 0 and irc == 0:
                # the index's bound per rank: what the FULLEST rank holds (the full-index partition builds over all ranks' ids with
                # W x this -- with this rank's own count a fuller neighbour made the bound too small; kssd_gpu_index_status says so
                # since the build stopped following a bound that is too small)
                st["idx_bound"] = min(cap, max(int(total), most if emu else 0) + 1024)
                break
            if rc not in (0, K.capi.ERR_OVERFLOW) or irc not in (0, K.capi.ERR_OVERFLOW):
                raise SystemExit("sketch / index failed: rc=%d, %d" % (rc, irc))
        else:
            raise SystemExit("sketch kept overflowing")
        sync()
        exchange_us = None
        if world > 1:
            # the exchange unit shrinks from the padded capacity to what the fullest rank really holds (+ a margin): setup,
            # untimed -- every step then gathers 20 % fewer bytes
            need = torch.tensor([st["idx_bound"]], dtype=torch.int64, device=dev)
            dist.all_reduce(need, op=dist.ReduceOp.MAX)
            st["unit"] = min(cap, (int(need.item()) + 4096 + 1023) // 1024 * 1024)
            st["idx_bound"] = min(cap, int(need.item()))   # (the fullest rank's count + 1 024, for every rank)
            new_search()
            index_build()
            rows()
            sync()
        unit = st["unit"]
        # setup, untimed like the sizing passes above: the clocks of an idle GPU need some tens of milliseconds of work to
        # settle (measured: the scan launch takes 0.54 ms in the first dozen steps after a pause and 0.52 ms from then on)
        # ... and a GPU that has just come out of other work (the test suite, a cold box) can sit in a slower state for seconds: the
        # spin-up goes on in groups of a.spinup steps until two groups in a row take the same time within 0.5 %, for at most ~2.5 s.
        # All ranks do the same number of groups (the decision is rank 0's time, broadcast by the all-reduce of `sync`-style state).
        spun = 0
        if a.spinup > 0:
            prev = None
            t_spin = time.perf_counter()
            while True:
                tg = time.perf_counter()
                run_steps(a.spinup)
                sync()
                tg = time.perf_counter() - tg
                spun += a.spinup
                stable = prev is not None and abs(tg - prev) <= 0.005 * prev
                stop = torch.tensor([1 if (stable or time.perf_counter() - t_spin > 2.5) else 0], dtype=torch.int32, device=dev)
                if world > 1:
                    dist.broadcast(stop, 0)
                if int(stop.item()):
                    break
                prev = tg
        spinup_done[0] = spun
        run_steps(a.warmup)
        sync()
        # at least ten launches of the timed region carry the events (the driver's 20 steps: every 2nd one)
        every = max(1, min(a.kernel_timing, a.steps // 10))
        ctx.kernel_time(0, reset=True)
        ctx.kernel_time(1, reset=True)
        ctx.set_kernel_timing(every)
        t0 = time.perf_counter()
        run_steps(a.steps)
        sync()
        dt = time.perf_counter() - t0
        rc, total, bad = ctx.sketch_status(stream)
        if rc != 0:
            raise SystemExit("sketch status rc=%d after the timed loop" % rc)
        scan_t, dist_t = ctx.kernel_times(0), ctx.kernel_times(1)
        scan_ms = float(scan_t.mean()) if len(scan_t) else 0.0   # average launch duration over every timed launch
        dist_ms = float(dist_t.mean()) if len(dist_t) else 0.0
        if ctx.index_status(stream) != 0:
            raise SystemExit("index status after the timed loop: the build overflowed")

        # The distance half alone, timed the same way (same warm-up, same number of steps, barrier + synchronize on both sides,
        # max over ranks): index build + the rank's rows on the sketches that are resident from the steps above -- what
        # `kssd dist -r` does once stage I is through (command_dist.c:763-790).  N > 1: the exchange is part of it.  Events
        # between the two halves of every pass: what the exchange + unpacking + index and what the rows cost on their own.
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(a.steps)]

        def run_dist(n, timed=False):
            for i in range(n):
                if timed:
                    ev[i][0].record(tstream)
                index_build()
                if timed:
                    ev[i][1].record(tstream)
                rows()
                if timed:
                    ev[i][2].record(tstream)
        run_dist(a.warmup)
        sync()
        ctx.kernel_time(1, reset=True)
        ctx.set_kernel_timing(every)
        t0 = time.perf_counter()
        run_dist(a.steps)
        sync()
        dt_dist = time.perf_counter() - t0
        dist_only_t = ctx.kernel_times(1)
        ctx.set_kernel_timing(0)
        run_dist(a.steps, timed=True)       # (a pass of its own: event records between the halves keep them from overlapping)
        sync()
        index_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in ev]))
        rows_ms = float(np.mean([e[1].elapsed_time(e[2]) for e in ev]))
        ctx.set_kernel_timing(1)
        if ctx.index_status(stream) != 0:
            raise SystemExit("index status after the distance loop: the build overflowed")
        tmax = torch.tensor([dt_dist], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt_dist = float(tmax.item())
        n_stage1, n_bloom = ctx.scan_stats(stream)

        dt_local = dt
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        ex_local = None

        if W > 1:
            # the exchange alone, untimed beside the steps: the two all-gathers (emulation: the copies of their bytes) + the
            # unpacking kernel of one step, between events on the stream they run on, averaged over 20 rounds behind a barrier
            g = st["search"].gather
            for _ in range(3):
                g(off_l, ids_l[:unit], stream=stream)
            sync()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(tstream)
            for _ in range(20):
                g(off_l, ids_l[:unit], stream=stream)
            e1.record(tstream)
            sync()
            ex_local = e0.elapsed_time(e1) * 1e3 / 20
            ex = torch.tensor([ex_local], dtype=torch.float64, device=dev)
            if world > 1:
                dist.all_reduce(ex, op=dist.ReduceOp.MAX)
            exchange_us = float(ex.item())
        per_rank = None
        if world > 1:
            # every rank's own figures side by side (the line's headline numbers are maxima over ranks): a measured scaling curve can be
            # read term by term against the one-GPU emulation of the same rank (profiles/r05n_bench_emu*.json)
            mine_v = torch.tensor([dt_local / a.steps * 1e3, scan_ms, index_ms, rows_ms, ex_local if ex_local is not None else 0.0, float(total)],
                                  dtype=torch.float64, device=dev)
            allv = torch.zeros(world * mine_v.numel(), dtype=torch.float64, device=dev)
            dist.all_gather_into_tensor(allv, mine_v)
            per_rank = [{"rank": r, "step_ms": v[0], "sketch_scan_ms": v[1], "index_ms": v[2], "rows_ms": v[3], "exchange_us": v[4], "ids": int(v[5])}
                        for r, v in enumerate(allv.view(world, -1).cpu().tolist())]

        szs = (off_l[1:] - off_l[:-1]).to(torch.int32)
        sh = shared.view(G, R)   # [this rank's genomes (its rows of the matrix)] x [all genomes], in either partition
        if world > 1:
            # size-independent check of the WHOLE distributed result, untimed: every rank's block gathered, the global matrix
            # assembled -- the ranks' rows one behind the other -- must be symmetric and carry every rank's sketch sizes on its
            # diagonal (any rank computing a wrong block shows: its block is the transpose of the others' columns)
            blocks = torch.zeros(world * G * R, dtype=torch.int32, device=dev)
            sizes_all = torch.zeros(world * G, dtype=torch.int32, device=dev)
            dist.all_gather_into_tensor(blocks, shared)
            dist.all_gather_into_tensor(sizes_all, szs.contiguous())
            full = blocks.view(world * G, R)
            assert torch.equal(full.diagonal(), sizes_all), "N > 1: diagonal of the global matrix != the sketch sizes"
            assert torch.equal(full, full.t()), "N > 1: the global all-pairs matrix is not symmetric"
            checksum = int(full.to(torch.int64).sum().item())          # the same number in either partition
            del blocks, full
        else:
            checksum = int(shared.to(torch.int64).sum().item())
        # size-independent sanity on this rank's block of the matrix
        own = sh[:, R_ * G:(R_ + 1) * G]
        assert torch.equal(own.diagonal(), szs), "diagonal of the all-pairs matrix must equal the sketch sizes"
        assert torch.equal(own, own.t()), "all-pairs shared-count matrix must be symmetric"
        if emu:
            # the foreign columns against the other ranks' sketch sizes: a genome shares at most min(|X|, |Y|) ids, and the emulated
            # ranks' batches are other clades: the block must not be all zero by accident of a wrong layout either -- column sums
            # of the own block are positive, every count is bounded by both sizes
            assert int(sh.max().item()) <= int(szs.max().item())
        res = dict(dt=dt, dt_dist=dt_dist, dist_only_ms=float(dist_only_t.mean()) if len(dist_only_t) else 0.0, dist_only_n=len(dist_only_t),
                   dist_only_t=dist_only_t, scan_ms=scan_ms, dist_ms=dist_ms, scan_t=scan_t, dist_t=dist_t, total=int(total), n_stage1=n_stage1,
                   n_bloom=n_bloom, exchange_us=exchange_us, unit=unit, checksum=checksum, index_ms=index_ms, rows_ms=rows_ms,
                   every=every, off=off_l.cpu().numpy(), ids=ids_l.cpu().numpy().view(np.uint32),
                   mask_lanes=None if summ is None else summary_clear_lanes(summ, int(chunk_off[-1])), per_rank=per_rank,
                   block=shared.cpu().numpy().view(np.uint32).reshape(G, R) if emu else None,
                   planes=[p.cpu().numpy().view(np.int64) for p in planes] if (emu and not a.no_planes and G * R <= 16_000_000) else None)
        ctx.close()
        st["search"] = None
        torch.cuda.empty_cache()
        return res

    PART_DESC = {"own": "own genomes indexed, all gathered sketches as query rows, the block written transposed: the rank's own rows, "
                        "row-major (all-pairs: queries = references, every metric symmetric; the index build stays constant per rank)",
                 "query": "full index of all gathered sketches on every rank, own query block as rows (north_star's literal partition; what "
                          "a search with Q != R needs)"}
    if W == 1:
        head_part, other_part = "query", None               # one GPU: both partitions are the same calls
    elif a.partition == "both":
        head_part, other_part = "own", "query"
    else:
        head_part, other_part = a.partition, None
    m = measure(head_part)
    other = measure(other_part) if other_part else None
    if other is not None:
        assert other["checksum"] == m["checksum"], "the two partitions compute different matrices"
        if emu:
            assert np.array_equal(other["block"], m["block"]), "the two partitions compute different blocks"
            if m["planes"] is not None:
                for x, y in zip(m["planes"], other["planes"]):
                    assert np.array_equal(x, y), "the two partitions compute different metric bits"
    dt, scan_ms, dist_ms, total = m["dt"], m["scan_ms"], m["dist_ms"], m["total"]
    tok = tokeniser_leg(a, shuf, local, dev, sink, m) if tok_leg else None

    if rank == 0:
        n_bases = G * L
        scan_bytes = 0.375 * n_bases + 4.0 * total        # 2-bit base + 1-bit mask, 4 B per id
        achieved = scan_bytes / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0
        # what the scan really asks the memory for (beside `traffic`, the counters' figure): with the batch's summary words the mask
        # stream shrinks to 8 B per chunk + the two words of every lane that holds a run-breaking position; `achieved` and `frac`
        # stay on SURVEY 8d's 0.375 B/base either way (they are a rate in the survey's unit, not a count of requests)
        n_chunks_b = int(chunk_off[-1])
        moved = (0.375 * n_chunks_b * 4096 if m["mask_lanes"] is None else 0.25 * n_chunks_b * 4096 + 8.0 * n_chunks_b + 8.0 * m["mask_lanes"]) + 4.0 * total
        pairs = G * R
        dist_bytes = (4 if a.no_planes else 36) * pairs + 4.0 * (total + R / G * total)
        spread = lambda t: {"min_ms": float(t.min()), "max_ms": float(t.max())} if len(t) else {}
        if W > 1:
            par = {"ranks": W, "backend": backend, "partition": head_part,
                   "what": "genomes sharded x%d for sketching (no communication); ONE exchange per step: all-gather of the packed "
                           "sketches (%s); every rank owns the rows of its own genomes (query-block sharding of the matrix); how it computes "
                           "them, '%s': %s"
                           % (W, "EMULATED on one GPU: device-to-device copies of the bytes, xGMI not included" if emu else
                              "torch.distributed backend '%s'%s" % (backend, " = RCCL over xGMI" if backend == "nccl" else " -- NOT RCCL: development check"),
                              head_part, PART_DESC[head_part])}
        else:
            par = "single GPU"
        res = {
            "metric": "genomes sketched/s (whole hot path per step: sketch + index + all-pairs distances, L3K10)",
            "value": world * G * a.steps / dt,
            "unit": "genomes/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": "BASELINE %s: %d synthetic %.1f Mb bacterial genomes per GPU (%d clades), "
                                   "L3K10 sketch + all-pairs" % (
                                       "configs[1]" if (G, L) == (1000, 5_000_000) else
                                       "configs[2] (the 500-clade generator stands in for GTDB r207: no data set on the box)" if G * world == 10000 else
                                       "configs[1]'s generator at another size", G, L / 1e6, a.clades),
                       "k": 10, "subk": 6, "drlevel": 3, "genomes_per_gpu": G, "genome_len": L,
                       "pairs_per_step": world * pairs,
                       "parallelism": par},
            "spinup": spinup_done[0],
            "pairs_per_s": world * pairs * a.steps / dt,
            "pairs_per_s_dist": world * pairs * a.steps / m["dt_dist"],
            "dist_ms_per_step": m["dt_dist"] / a.steps * 1e3,
            "dist_halves_ms": {"exchange_unpack_index": m["index_ms"], "rows": m["rows_ms"],
                               "what": "the distance half's two parts between events on the step's stream (a pass of their own: the "
                                       "records keep the parts from overlapping)"},
            "mbase_per_s": world * n_bases * a.steps / dt / 1e6,
            "ids_per_batch": int(total),
            "kernels": {"sketch_scan_ms": scan_ms, "dist_rows_ms": dist_ms, "launches_timed": [len(m["scan_t"]), len(m["dist_t"])],
                        "sketch_scan_spread": spread(m["scan_t"]), "dist_rows_spread": spread(m["dist_t"]),
                        "timed_every": m["every"],   # every N-th launch of the timed region carries the events (--kernel-timing)
                        "dist_rows_GBs": dist_bytes / (dist_ms * 1e-3) / 1e9 if dist_ms > 0 else None,
                        "scan_positions_past_stage1": m["n_stage1"] / n_bases, "scan_positions_past_bloom": m["n_bloom"] / n_bases},
            "roofline": {"bound": "hbm", "kernel": "sketch_scan_kernel<6>", "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": None, "traffic_source": None,
                         "bytes_moved": moved, "bytes_moved_what": (
                             "requested per launch: packed bases 0.25 B/position incl. padding + 8 B of summary per chunk + 8 B of mask for "
                             "each of the %d lanes (of %d) whose 64 positions are not all bases + 4 B per id" % (m["mask_lanes"], n_chunks_b * 64)
                             if m["mask_lanes"] is not None else "requested per launch: the whole packed and mask streams incl. padding + 4 B per id (--no-mask-summary)"),
                         "algorithmic_bytes_per_launch": scan_bytes, "launches_timed": len(m["scan_t"]), **spread(m["scan_t"])},
            # the distance half as its own quantity (BASELINE's metric names two rates): `pairs_per_s_dist` above is its
            # whole-job rate (index build + rows, `steps` timed passes), this is its dominant kernel against the same roofline:
            # 36 B per pair written (4 B shared count + four f64 metrics) + 4 B per query and reference id read
            "roofline_dist": {"bound": "hbm", "kernel": "dist_rows_kernel",
                              "achieved": dist_bytes / (m["dist_only_ms"] * 1e-3) / 1e9 if m["dist_only_ms"] > 0 else 0.0,
                              "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": dist_bytes / (m["dist_only_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS if m["dist_only_ms"] > 0 else 0.0,
                              "traffic": None, "traffic_source": None,
                              "algorithmic_bytes_per_launch": dist_bytes, "launch_ms": m["dist_only_ms"],
                              "launches_timed": m["dist_only_n"], **spread(m["dist_only_t"])},
        }
        if tok is not None:
            res["roofline_tok"] = tok
        if W > 1:
            res["backend"] = backend
            res["ranks_seen"] = world
            res["exchange"] = {"us": m["exchange_us"], "unit_ids_per_rank": m["unit"], "bytes_gathered_per_rank": W * (4 * m["unit"] + 8 * (G + 1)),
                               "what": ("device-to-device copies of the bytes the two all-gathers deliver + the unpacking kernel (xGMI NOT included)" if emu else
                                        "two all_gather_into_tensor (offsets, padded id units) + the unpacking kernel") +
                                       ", alone on the step's stream, mean of 20, max over ranks"}
            res["matrix_checksum"] = m["checksum"]
            if m["per_rank"] is not None:
                res["per_rank"] = m["per_rank"]     # step / scan / exchange+unpack+index / rows / exchange alone, rank by rank (ms; exchange in us)
            if not emu:
                res["runtime"] = {"hip": K.gpu_lib().kssd_gpu_runtime_path(0).decode(), "mapped": K.capi.runtime_paths(),
                                  "what": "the HIP runtime libkssd_gpu.so is bound to and every HIP / RCCL / HSA file mapped into rank 0 (one of each: "
                                          "public_kssd_amd.capi.assert_single_runtime)"}
            if W > 1 and head_part == "own":
                del res["roofline_dist"]      # (the rows kernel of this partition writes counts only; its metrics leave by the transposing kernel)
            if other is not None:
                res["partition_" + other_part] = {
                    "value": world * G * a.steps / other["dt"], "unit": "genomes/s", "ms_per_step": other["dt"] / a.steps * 1e3,
                    "pairs_per_s": world * pairs * a.steps / other["dt"], "exchange_us": other["exchange_us"],
                    "sketch_scan_ms": other["scan_ms"], "dist_rows_ms": other["dist_ms"],
                    "dist_ms_per_step": other["dt_dist"] / a.steps * 1e3,
                    "dist_halves_ms": {"exchange_unpack_index": other["index_ms"], "rows": other["rows_ms"]},
                    "what": PART_DESC[other_part] + "; same steps / warmup / batch, measured after the headline; the same rows of the "
                            "matrix, the same checksum"}
        if emu:
            # one GPU played one rank: the line is a per-rank cost, not a job's throughput -- `value` is what ONE such rank sketches
            # per second, and says so
            res["emulated"] = {"world": W, "rank": R_, "per_rank_ms": dt / a.steps * 1e3, "index_ms": m["index_ms"], "rows_ms": m["rows_ms"],
                               "exchange_bytes": W * (4 * m["unit"] + 8 * (G + 1)), "exchange_copy_us": m["exchange_us"],
                               "partition": head_part,
                               "what": "ONE GPU as rank %d of %d at the per-GPU size: own batch sketched every step, the other ranks' units "
                                       "(sketched once, untimed) delivered by device-to-device copies of exactly the bytes the all-gather "
                                       "delivers, then unpacking + index + the rank's rows (%d x %d pairs, 36 B each); xGMI NOT included; "
                                       "n_gpus stays 1 and `value` is this one rank's genomes/s" % (R_, W, G, R)}
            if other is not None:
                res["emulated"]["partition_" + other_part] = {"per_rank_ms": other["dt"] / a.steps * 1e3, "index_ms": other["index_ms"],
                                                             "rows_ms": other["rows_ms"]}
                res["emulated"]["blocks_identical"] = "shared counts%s of the two partitions' blocks are bit-identical" % (
                    " and all four metric planes" if m["planes"] is not None else "")
        if a.cpu_sample and W == 1 and kept:
            ol, il = m["off"], m["ids"]
            gpu_sets = [il[int(ol[g]):int(ol[g + 1])] for g in range(len(kept))]
            cores = host_cores()
            cb = cpu_baseline(shuf, kept, cores, gpu_sets, a.e2e_files, e2e_search=a.e2e_search, e2e_search4k=a.e2e_search_large,
                              e2e_sketch_large=a.e2e_sketch_large)
            res["cpu_baseline"] = cb.get("reference", cb["port"])
            res["cpu_baseline_port"] = cb["port"]
            res["cpu_baseline_dist"] = cb.get("dist_reference", cb["dist_port"])
            res["cpu_baseline_dist_port"] = cb["dist_port"]
            if "end_to_end" in cb:
                res["end_to_end"] = cb["end_to_end"]
            if "reference_gz" in cb:
                res["cpu_baseline_gz"] = cb["reference_gz"]
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc) and G == 1000 and L == 5_000_000 and W == 1:  # the PMC passes were collected on the default workload
            # counters cannot be collected inside a timed run: the figures are RECORDED by profiles/pmc_refresh.py, which stamps
            # them with a hash of the kernel sources they were measured on -- sources that have changed since: no figure
            try:
                pj = json.load(open(pmc))
                if pj.get("source_sha") == K.capi.kernel_source_sha():
                    src = ("RECORDED, not measured by this run: profiles/pmc_traffic.json (rocprofv3 --pmc passes of %s, "
                           "(FETCH_SIZE x 2 + WRITE_SIZE) x 1024 per launch, kernel sources %s)" % (pj.get("tag", "?"), pj["source_sha"][:12]))
                    res["roofline"]["traffic"] = pj.get("sketch_scan_bytes_per_launch")
                    res["roofline"]["traffic_source"] = src
                    res["roofline_dist"]["traffic"] = pj.get("dist_rows_bytes_per_launch")
                    res["roofline_dist"]["traffic_source"] = src
                    if "roofline_tok" in res:
                        res["roofline_tok"]["traffic"] = pj.get("tok_bytes_per_launch")
                        res["roofline_tok"]["traffic_source"] = src
                else:
                    res["roofline"]["traffic_source"] = res["roofline_dist"]["traffic_source"] = (
                        "none: profiles/pmc_traffic.json was recorded on other kernel sources (%s, now %s): run profiles/pmc_refresh.py"
                        % (str(pj.get("source_sha"))[:12], K.capi.kernel_source_sha()[:12]))
            except Exception:
                pass
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main() or 0)
