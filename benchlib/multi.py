"""The N > 1 plumbing of bench.py that measures nothing by itself and touches no oracle: the step on N devices driven by ONE process
(`--exchange c`: kssd_gpu_allgather_sketches, no torch.distributed) and the exchange of ONE rank of N played on one GPU
(`--emulate-world`)."""
import json
import time

import numpy as np
import torch

import public_kssd_amd as K
from benchlib.launch import log
from benchlib.workloads import make_batch, mask_summary

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def run_exchange_c(a, shuf, n_dev, reason=None):
    """`--exchange c`: the same step on N devices driven by ONE process -- no torch.distributed anywhere: every device sketches
    its own batch, ONE kssd_gpu_allgather_sketches (RCCL inside libkssd_gpu.so: ncclCommInitAll + grouped ncclAllGather) hands
    every device all sketches, every device builds the full index and computes its own block of query rows (north_star's
    partition).  K steps between two synchronisations of all devices; one JSON line.  The second, independent way for an N > 1
    run to succeed (the first: one process per GPU under torch.distributed.run, ShardedSearch)."""
    G, L = a.genomes, a.length
    exp_ids = int(G * L / 4096)
    cap = int(exp_ids * 1.25) + 4096
    R = G * n_dev

    class Dev:
        pass
    dv = []
    t0 = time.time()
    for d in range(n_dev):
        v = Dev()
        torch.cuda.set_device(d)
        v.dev = torch.device("cuda", d)
        v.packed, v.mask, v.chunk_off, _ = make_batch(G, L, a.clades, 20260101 + 7919 * d, v.dev)
        v.ctx = K.GpuCtx(shuf, d)
        v.summ = None if getattr(a, "no_mask_summary", False) else mask_summary(v.ctx, v.mask, int(v.chunk_off[-1]), v.dev)
        v.off_l = torch.zeros(G + 1, dtype=torch.int64, device=v.dev)
        v.ids_l = torch.zeros(cap, dtype=torch.int32, device=v.dev)
        v.roff = torch.zeros(R + 1, dtype=torch.int64, device=v.dev)
        v.rids = torch.zeros(n_dev * cap, dtype=torch.int32, device=v.dev)
        v.shared = torch.zeros(G * R, dtype=torch.int32, device=v.dev)
        v.planes = [None] * 4 if a.no_planes else [torch.zeros(G * R, dtype=torch.float64, device=v.dev) for _ in range(4)]
        v.tstream = torch.cuda.Stream(device=v.dev)
        v.stream = v.tstream.cuda_stream
        dv.append(v)
    for v in dv:
        torch.cuda.synchronize(v.dev)
    log("[bench] --exchange c: %d devices, %d x %.1f Mb each, batches packed in %.1f s" % (n_dev, G, L / 1e6, time.time() - t0))
    unit = [cap]
    bound = [cap]

    def sync_all():
        for v in dv:
            torch.cuda.synchronize(v.dev)

    def steps(n):
        for _ in range(n):
            for v in dv:
                v.ctx.sketch_plan(v.packed, v.mask, v.chunk_off, v.off_l, v.ids_l, cap, K.SKETCH_FASTA, 1, d_summary=v.summ)
                for ph in (K.PHASE_PREP, K.PHASE_SCAN, K.PHASE_EXACT, K.PHASE_FINISH):
                    v.ctx.sketch_phase(ph, v.stream)
            K.GpuCtx.allgather_sketches([v.ctx for v in dv], [v.off_l for v in dv], [v.ids_l for v in dv], G, unit[0],
                                        [v.roff for v in dv], [v.rids for v in dv], streams=[v.stream for v in dv])
            for v in dv:
                v.ctx.index_build_device(v.roff, v.rids, R, bound[0] * n_dev, v.stream, check=False)  # (no status read-back inside the loop: the devices must not wait for one another on the host)
                v.ctx.dist_device(v.off_l, v.ids_l, G, 0, G, v.shared, *v.planes, stream=v.stream)

    for attempt in range(6):          # sizing passes (workspaces, overflow retries), untimed
        try:
            steps(1)
            sync_all()
        except K.KssdError as e:
            # the first collective of the run: librccl.so, ncclCommInitAll over the device list, peer access, the grouped all-gathers.
            # ONE line with the reason, a non-zero exit, nothing retried inside this process (a communicator that failed to come up
            # is not torn down and built again under a timed run)
            print(json.dumps({"error": "exchange failed", "where": "kssd_gpu_allgather_sketches (RCCL inside libkssd_gpu.so) on %d devices" % n_dev,
                              "code": e.code, "reason": str(e)[:600], "hip": K.gpu_lib().kssd_gpu_last_hip_error().decode(errors="replace")[:300]}),
                  file=__import__("sys").stderr, flush=True)
            raise SystemExit(3)
        rcs = [v.ctx.sketch_status(v.stream) for v in dv]
        ircs = [v.ctx.index_status(v.stream) for v in dv]
        if all(r[0] == 0 for r in rcs) and all(i == 0 for i in ircs):
            break
        if any(r[0] not in (0, K.capi.ERR_OVERFLOW) for r in rcs) or any(i not in (0, K.capi.ERR_OVERFLOW) for i in ircs):
            raise SystemExit("sketch / index failed: %s %s" % (rcs, ircs))
    else:
        raise SystemExit("sketch kept overflowing")
    most = max(int(r[1]) for r in rcs)
    unit[0] = min(cap, (most + 4096 + 1023) // 1024 * 1024)   # the exchange unit: what the fullest device really holds
    bound[0] = min(cap, most + 1024)
    steps(2)
    sync_all()
    if a.spinup > 0:
        steps(a.spinup)
        sync_all()
    steps(a.warmup)
    sync_all()
    for v in dv:
        v.ctx.kernel_time(0, reset=True)
        v.ctx.kernel_time(1, reset=True)
    t0 = time.perf_counter()
    steps(a.steps)
    sync_all()
    dt = time.perf_counter() - t0
    scan = [v.ctx.kernel_time(0) for v in dv]
    rows = [v.ctx.kernel_time(1) for v in dv]
    for v in dv:
        if v.ctx.sketch_status(v.stream)[0] != 0 or v.ctx.index_status(v.stream) != 0:
            raise SystemExit("status after the timed loop")
    # the exchange alone (two grouped all-gathers + unpacking on every device), mean of 20, host-timed between synchronisations
    ex = lambda: K.GpuCtx.allgather_sketches([v.ctx for v in dv], [v.off_l for v in dv], [v.ids_l for v in dv], G, unit[0],
                                             [v.roff for v in dv], [v.rids for v in dv], streams=[v.stream for v in dv])
    for _ in range(3):
        ex()
    sync_all()
    te = time.perf_counter()
    for _ in range(20):
        ex()
    sync_all()
    exchange_us = (time.perf_counter() - te) / 20 * 1e6
    # the whole result, assembled: symmetric, every device's sketch sizes on the diagonal
    full = torch.cat([v.shared.view(G, R).cpu() for v in dv], 0)
    sizes = torch.cat([(v.off_l[1:] - v.off_l[:-1]).to(torch.int32).cpu() for v in dv])
    assert torch.equal(full.diagonal(), sizes), "diagonal of the global matrix != the sketch sizes"
    assert torch.equal(full, full.t()), "the global all-pairs matrix is not symmetric"
    total = int(dv[0].off_l[-1].item())
    scan_ms = sum(m * n for m, n in scan) / max(1, sum(n for _, n in scan))
    rows_ms = sum(m * n for m, n in rows) / max(1, sum(n for _, n in rows))
    n_bases = G * L
    scan_bytes = 0.375 * n_bases + 4.0 * total
    achieved = scan_bytes / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0
    lib = K.gpu_lib()
    res = {
        "metric": "genomes sketched/s (whole hot path per step: sketch + index + all-pairs distances, L3K10)",
        "value": n_dev * G * a.steps / dt, "unit": "genomes/s", "n_gpus": n_dev, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64",
        "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: %d synthetic %.1f Mb bacterial genomes per GPU (%d clades), L3K10 sketch + all-pairs"
                               % (G, L / 1e6, a.clades),
                   "k": 10, "subk": 6, "drlevel": 3, "genomes_per_gpu": G, "genome_len": L, "pairs_per_step": n_dev * G * R,
                   "parallelism": {"ranks": n_dev, "processes": 1, "backend": "rccl (kssd_gpu_allgather_sketches inside one process, no torch.distributed)",
                                   "partition": "query", "what": "genomes sharded x%d for sketching; ONE exchange per step: grouped ncclAllGather of the "
                                   "packed sketches; full index on every device, own query block as rows" % n_dev}},
        "spinup": a.spinup, "pairs_per_s": n_dev * G * R * a.steps / dt, "ids_per_batch": total,
        "kernels": {"sketch_scan_ms": scan_ms, "dist_rows_ms": rows_ms},
        "per_rank": [{"rank": d, "device": d, "sketch_scan_ms": scan[d][0], "rows_ms": rows[d][0], "ids": int(dv[d].off_l[-1].item())} for d in range(n_dev)],
        "roofline": {"bound": "hbm", "kernel": "sketch_scan_kernel<6>", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes_per_launch": scan_bytes},
        "exchange": {"kind": "c", "us": exchange_us, "unit_ids_per_rank": unit[0],
                     "what": "kssd_gpu_allgather_sketches alone (offsets + padded id units, unpacking kernel on every device), mean of 20, host clock "
                             "between synchronisations of all devices"},
        "runtime": {"hip": lib.kssd_gpu_runtime_path(0).decode(), "rccl": lib.kssd_gpu_runtime_path(1).decode(), "mapped": K.capi.runtime_paths()},
        "matrix_checksum": int(full.to(torch.int64).sum().item()),
    }
    if reason:
        res["exchange"]["fallback_from_torch_distributed"] = reason
    print(json.dumps(res), flush=True)
    for v in dv:
        v.ctx.close()
    return 0


class EmulatedGather:
    """The exchange of ONE rank of `world` played on one GPU (bench.py --emulate-world): the other ranks' units were sketched
    once, untimed, from their own batches; every step they are delivered by device-to-device copies of exactly the bytes the
    all-gather delivers into this rank's staging -- world x (8 (G + 1) + 4 unit) bytes, the own unit fresh from this step's
    sketch --, then the same unpacking kernel (kssd_gpu_concat_units_device).  What it leaves out is xGMI: the line says so."""

    def __init__(self, world, rank, G, unit, dev, engine, units):
        self.world, self.rank, self.G, self.cap, self.engine = world, rank, G, unit, engine
        self.src_off = torch.zeros(world * (G + 1), dtype=torch.int64, device=dev)
        self.src_ids = torch.zeros(world * unit, dtype=torch.int32, device=dev)
        for j, (off_j, ids_j) in units.items():
            self.src_off[j * (G + 1):(j + 1) * (G + 1)] = off_j
            n = min(unit, int(ids_j.numel()))
            self.src_ids[j * unit:j * unit + n] = ids_j[:n]
        self.off_all = torch.zeros(world * (G + 1), dtype=torch.int64, device=dev)
        self.ids_all = torch.zeros(world * unit, dtype=torch.int32, device=dev)
        self.roff = torch.zeros(world * G + 1, dtype=torch.int64, device=dev)
        self.rids = torch.zeros(world * unit, dtype=torch.int32, device=dev)
        self.bytes = world * (8 * (G + 1) + 4 * unit)

    def __call__(self, off_l, ids_l, group=None, stream=None):
        w, r, G, u = self.world, self.rank, self.G, self.cap
        for dst, src, own, n in ((self.off_all, self.src_off, off_l, G + 1), (self.ids_all, self.src_ids, ids_l, u)):
            if r > 0:
                dst[:r * n].copy_(src[:r * n], non_blocking=True)
            dst[r * n:(r + 1) * n].copy_(own[:n], non_blocking=True)
            if r + 1 < w:
                dst[(r + 1) * n:].copy_(src[(r + 1) * n:], non_blocking=True)
        self.engine.concat_units_device(self.off_all, self.ids_all, w, G, u, self.roff, self.rids, stream)
        return self.roff, self.rids
