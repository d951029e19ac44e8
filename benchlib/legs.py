"""The legs beside bench.py's headline loop: the CPU comparators with the end-to-end commands (`cpu_baseline`: the oracle, the reference
binary and this build's command line on the same files, every result cross-checked while it is measured), the device tokeniser's leg
(`tokeniser_leg`: roofline_tok), and the two workloads that are not the headline (`run_fastq`: BASELINE configs[3], `run_mammal`:
configs[4]) with their parity checks.  Everything here runs OUTSIDE the headline's timed region; the oracle and oracle/_ref are used
only as the CPU baseline and as the checker of results (never as the thing measured for `value`)."""
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch

import public_kssd_amd as K
from benchlib.launch import host_cores, log
from benchlib.workloads import READ_LEN, make_batch, make_long_records, make_reads_batch, mask_summary, summary_clear_lanes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# (the synthetic workloads -- make_batch, make_reads_batch, make_long_records -- and the launcher plumbing live in benchlib/)


# ------------------------------------------------------------------------------------------------------
# CPU baseline + end-to-end leg (rank 0, N=1, bounded sample, outside the timed region)
# ------------------------------------------------------------------------------------------------------
KSSD_BIN = os.path.join(ROOT, "public_kssd_amd", "kssd")


def _run_ours(args, cwd, env_extra=None, timeout=900):
    env = dict(os.environ)
    env.update(env_extra or {})
    t0 = time.time()
    r = subprocess.run([KSSD_BIN] + [str(a) for a in args], cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout, env=env)
    dt = time.time() - t0
    if r.returncode != 0:
        raise RuntimeError("kssd %s -> %d\n%s" % (args, r.returncode, r.stderr.decode(errors="replace")[-2000:]))
    timing = [json.loads(line) for line in r.stderr.decode(errors="replace").splitlines() if line.startswith('{"kssd_timing"')]
    # (one stage line per command; the one-command all-pairs flow prints two: keyed by their names then)
    return dt, (None if not timing else timing[0] if len(timing) == 1 else {t["kssd_timing"]: t for t in timing})


def cpu_baseline(shuf, kept, cores, gpu_sets, e2e_files, gz_distinct=128, e2e_search=1024, e2e_search4k=4096, e2e_sketch_large=8192):
    """The CPU comparators and the end-to-end leg, on the same inputs in the same run:
      port       oracle/kssd_oracle.c (our restatement) sketching the sample texts, OpenMP over genomes
      reference  oracle/_ref/kssd (the real reference, when the snapshot carries it): stage I on FASTA files in tmpfs,
                 stage II (its fixed-cost 2 GiB mco.index), search incl. distance.out
      end_to_end the product's own command line on the same files: read + tokenise + H2D + kernels + D2H + file write
    The GPU sketches of the sample are checked against the oracle and the reference while we are here."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import kssd_oracle as ko
    from synth import fasta_text
    texts = [fasta_text(c, b"g%d" % i, n_mask=m) for i, (c, m) in enumerate(kept)]
    nb = sum(len(c) for c, _ in kept)
    n = len(texts)
    out = {}
    # the port (our C restatement of the reference algorithm), OpenMP over genomes like run_stageI
    t0 = time.time()
    off, ids = ko.sketch_texts(shuf.table, shuf.k, shuf.subk, shuf.drlevel, texts, threads=cores)
    t_port = time.time() - t0
    for g in range(n):
        want = np.sort(ids[int(off[g]):int(off[g + 1])])
        assert np.array_equal(gpu_sets[g], want), "bench sample genome %d: GPU sketch != oracle" % g
    out["port"] = {"value": n / t_port, "unit": "genomes/s", "cores": min(cores, n), "kind": "port",
                   "sample": "%d of the bench genomes (%.0f Mbase) as 70-col FASTA text in memory, oracle/kssd_oracle.c "
                             "sketch_texts, OpenMP over genomes" % (n, nb / 1e6),
                   "mbase_per_s": nb / 1e6 / t_port}
    t0 = time.time()
    sh = ko.shared_counts(off, ids, off, ids, threads=cores)
    t_cnt = time.time() - t0
    out["dist_port"] = {"value": sh.size / t_cnt, "unit": "pairs/s", "cores": cores, "kind": "port",
                        "sample": "%dx%d all-pairs of the sample sketches, index build + posting traversal only" % (n, n)}
    d = tempfile.mkdtemp(prefix="kssd_bench_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        os.mkdir(os.path.join(d, "fa"))
        reps = max(1, e2e_files // n)
        nf = 0
        for i, t in enumerate(texts):
            with open(os.path.join(d, "fa", "r00_g%04d.fasta" % i), "wb") as f:
                f.write(t)
            nf += 1
            for r in range(1, reps):   # the same genomes again under other names: whole-pipeline work, bounded setup time
                os.symlink("r00_g%04d.fasta" % i, os.path.join(d, "fa", "r%02d_g%04d.fasta" % (r, i)))
                nf += 1
        # the search legs run on the first ns names (an all-pairs report is 114 bytes a pair: 10 000 x 10 000 would be 11 GB of text)
        ns = min(nf, max(2, e2e_search))
        sub = nf != ns
        if sub:
            os.mkdir(os.path.join(d, "fa_s"))
            for nm in sorted(os.listdir(os.path.join(d, "fa")))[:ns]:
                os.symlink(os.path.join("..", "fa", nm), os.path.join(d, "fa_s", nm))
        # the same genomes gzip'ed (level 1: the setup stays in seconds), under as many names: zlib on our host threads against the
        # reference's `zcat -fc` pipes
        import zlib
        os.mkdir(os.path.join(d, "gz"))
        n_gz = min(n, gz_distinct)
        gz_reps = max(1, min(reps, 8))
        ngz = 0
        for i in range(n_gz):
            co = zlib.compressobj(1, zlib.DEFLATED, 31)
            with open(os.path.join(d, "gz", "r00_g%04d.fasta.gz" % i), "wb") as f:
                f.write(co.compress(texts[i]) + co.flush())
            ngz += 1
            for r in range(1, gz_reps):
                os.symlink("r00_g%04d.fasta.gz" % i, os.path.join(d, "gz", "r%02d_g%04d.fasta.gz" % (r, i)))
                ngz += 1
        del texts
        shuf.write(os.path.join(d, "L3K10.shuf"))
        fa_desc = "%d FASTA files in tmpfs (%d distinct bench genomes of %.1f Mb x %d names, %.0f Mbase)" % (nf, n, nb / n / 1e6, reps, nb * reps / 1e6)
        # ---- end to end through the product's command line ----
        if os.access(KSSD_BIN, os.X_OK):
            # twice: HIP initialisation and context creation vary by ~0.1 s from process to process (context creation 0.05 -
            # 0.19 s, profiles/r02B_stream_probe.txt), which says nothing about the path; both times are reported
            dt_first, tm_first = _run_ours(["dist", "-p", cores, "-L", "L3K10.shuf", "-o", "our_sk0", "fa"], d, {"KSSD_TIMING": "1"})
            dt, tm = _run_ours(["dist", "-p", cores, "-L", "L3K10.shuf", "-o", "our_sk", "fa"], d, {"KSSD_TIMING": "1"})
            runs = [dt_first, dt]
            if dt_first < dt:
                dt, tm = dt_first, tm_first
            ours = ko.sketch_sets_by_name(os.path.join(d, "our_sk"))
            for i in range(n):
                assert np.array_equal(ours["r00_g%04d.fasta" % i], gpu_sets[i]), "kssd CLI sketch != device-level sketch"
            e2e = {"value": nf / dt, "unit": "genomes/s", "mbase_per_s": nb * reps / 1e6 / dt, "seconds": dt, "host_threads": cores,
                   "what": "`kssd dist -L L3K10.shuf -o <dir> <fasta dir>`: process start, .shuf load, files read into page-locked "
                           "buffers on the host threads, raw text H2D, tokenised on the device, sketch kernels, D2H, slot order, "
                           "combco.* written -- wall time of the command, the better of two runs (HIP start-up varies by ~0.1 s from "
                           "process to process)", "sample": fa_desc, "stages": tm, "seconds_runs": runs}
            # ---- the same command where start-up is amortised: stage I alone on 8 192 names hard-linked onto the distinct files ----
            n_big = (e2e_sketch_large // n) * n if e2e_sketch_large >= 4 * nf else 0
            if n_big:
                os.mkdir(os.path.join(d, "fa_big"))
                for r in range(n_big // n):
                    for i in range(n):
                        os.link(os.path.join(d, "fa", "r00_g%04d.fasta" % i), os.path.join(d, "fa_big", "h%03d_g%04d.fasta" % (r, i)))
                dtb, tmb = _run_ours(["dist", "-p", cores, "-L", "L3K10.shuf", "-o", "our_sk_big", "fa_big"], d, {"KSSD_TIMING": "1"})
                ob = ko.sketch_sets_by_name(os.path.join(d, "our_sk_big"))
                for i in (0, n // 2, n - 1):
                    for r in (0, n_big // n - 1):
                        assert np.array_equal(ob["h%03d_g%04d.fasta" % (r, i)], gpu_sets[i]), "kssd CLI sketch (8 192-file leg) != device-level sketch"
                shutil.rmtree(os.path.join(d, "our_sk_big"), ignore_errors=True)
                shutil.rmtree(os.path.join(d, "fa_big"), ignore_errors=True)
                st1 = (tmb if "s_total" in tmb else tmb.get("stage1")) if tmb else None   # (one timing line, or several keyed by their names)
                big = {"value": n_big / dtb, "unit": "genomes/s", "mbase_per_s": nb / n * n_big / 1e6 / dtb, "seconds": dtb, "host_threads": cores, "stages": tmb,
                       "what": "`kssd dist -L L3K10.shuf -o <dir> <fasta dir>`, stage I only, on %d names hard-linked onto the %d distinct files in tmpfs: "
                               "the same command as `end_to_end` with process start, hipInit and the first context amortised; wall time of one run" % (n_big, n)}
                if st1:
                    jobs = max(1, int(st1["batches"]))
                    startup = float(st1["s_context_create_max"]) + float(st1["s_before_workers"])
                    steady = max(1e-9, float(st1["s_total"]) - startup - float(st1["s_assemble_write"]))
                    per = {"wall_ms_per_job_steady": steady / jobs * 1e3,
                           "h2d_floor_ms_per_job": float(st1["text_bytes"]) / jobs / 53e9 * 1e3,
                           "reader_threads_ms_per_job": (float(st1["s_copy_threads_summed"]) + float(st1["s_unpack_threads_summed"])) / max(1, int(st1["host_threads"])) / jobs * 1e3,
                           "device_calls_ms_per_job_per_worker": float(st1["s_device_calls_summed"]) / jobs / max(1, int(st1.get("workers", 2 * max(1, int(st1["gpus"]))))) * 1e3,
                           "workers": int(st1.get("workers", 0))}
                    names = {"h2d_floor_ms_per_job": "PCIe (H2D of the raw text at ~53 GB/s)", "reader_threads_ms_per_job": "the readers (read(2) out of tmpfs into the jobs' texts, summed thread time / threads)",
                             "device_calls_ms_per_job_per_worker": "the device calls (H2D + kernels + D2H as the workers see them, summed / workers)"}
                    top = max(names, key=lambda k_: per[k_])
                    per["steady_genomes_per_s"] = n_big / steady
                    per["bound"] = ("%s: %.2f ms of the %.2f ms a job takes in the steady state" % (names[top], per[top], per["wall_ms_per_job_steady"])
                                    + ("" if per[top] > 0.7 * per["wall_ms_per_job_steady"] else
                                       " -- none of the three parts fills the interval: the rest is hand-over between readers, workers and the device (waves of files per job)"))
                    big["per_job"] = per
                e2e["sketch_%d" % n_big] = big
            sk_s = "our_sk"
            dt_s = dt
            if sub:
                _run_ours(["dist", "-p", cores, "-L", "L3K10.shuf", "-o", "our_sk_s", "fa_s"], d)
                sk_s = "our_sk_s"
            dt2a, tm2a = _run_ours(["dist", "-p", cores, "-r", sk_s, "-o", "our_dist0", sk_s], d, {"KSSD_TIMING": "1"})
            dt2, tm2 = _run_ours(["dist", "-p", cores, "-r", sk_s, "-o", "our_dist", sk_s], d, {"KSSD_TIMING": "1"})
            runs2 = [dt2a, dt2]
            if dt2a < dt2:
                dt2, tm2 = dt2a, tm2a
            e2e["search"] = {"value": ns * ns / dt2, "unit": "pairs/s", "seconds": dt2, "seconds_runs": runs2, "stages": tm2,
                             "what": "`kssd dist -r <sketches> -o <dir> <sketches>`: %d x %d all-pairs incl. process start, reading the "
                                     "sketches, the device search and the distance.out text (%d MB) on %d host threads -- the same command "
                                     "line the reference is timed with below; wall time of the command, the better of two runs"
                                     % (ns, ns, os.path.getsize(os.path.join(d, "our_dist", "distance.out")) >> 20, cores)}
            # the one-command flow: stage I, ONE exchange (a one-rank RCCL communicator here), index and rows on the sketches the
            # device still holds, distance.out -- against the two commands above (sketch, then search) on the same files
            fa_ap = "fa_s" if sub else "fa"
            dta0, _ = _run_ours(["dist", "-p", cores, "-L", "L3K10.shuf", "-o", "our_ap0", "--allpairs", fa_ap], d)
            dta, tma = _run_ours(["dist", "-p", cores, "-L", "L3K10.shuf", "-o", "our_ap", "--allpairs", "--keepskf", fa_ap], d, {"KSSD_TIMING": "1"})
            _run_ours(["dist", "-p", cores, "-r", sk_s, "-o", "our_dist_k", "--keepskf", sk_s], d)
            for fn in ("sharedk_ct.dat", "distance.out"):
                assert open(os.path.join(d, "our_ap", fn), "rb").read() == open(os.path.join(d, "our_dist_k", fn), "rb").read(), \
                    "kssd dist --allpairs: %s differs from the two-command flow's" % fn
            e2e["allpairs"] = {"value": ns / min(dta, dta0), "unit": "genomes/s", "pairs_per_s": ns * ns / min(dta, dta0), "seconds_runs": [dta0, dta],
                               "stages": tma,
                               "what": "`kssd dist -L L3K10.shuf -o <dir> --allpairs <fasta dir>` on %d files: sketch + all-pairs + distance.out in ONE "
                                       "command, the sketches never leave the device (kssd_gpu_resident_*, one RCCL all-gather); sharedk_ct.dat and "
                                       "distance.out byte-identical to the two-command flow's; wall time, the better of two runs" % ns,
                               "two_commands_seconds": dt_s + dt2 if not sub else None}
            if ngz:
                dtg0, _ = _run_ours(["dist", "-p", cores, "-L", "L3K10.shuf", "-o", "our_gz0", "gz"], d)
                dtg, tmg = _run_ours(["dist", "-p", cores, "-L", "L3K10.shuf", "-o", "our_gz", "gz"], d, {"KSSD_TIMING": "1"})
                og = ko.sketch_sets_by_name(os.path.join(d, "our_gz"))
                for i in range(n_gz):
                    assert np.array_equal(og["r00_g%04d.fasta.gz" % i], gpu_sets[i]), "kssd CLI sketch of the gzip'ed genome != device-level sketch"
                e2e["gzip"] = {"value": ngz / min(dtg, dtg0), "unit": "genomes/s", "seconds_runs": [dtg0, dtg], "stages": tmg,
                               "sample": "%d .fasta.gz files in tmpfs (%d distinct genomes x %d names, gzip level 1)" % (ngz, n_gz, gz_reps)}
            out["end_to_end"] = e2e
        # ---- the real reference binary when the snapshot carries it ----
        if ko.have_ref() and shutil.which("zcat"):
            # the reference only goes parallel when there are more files than threads (command_dist.c:275)
            p_ref = max(1, min(cores, nf - 1))
            t0 = time.time()
            ko.run_ref(["dist", "-p", p_ref, "-L", "L3K10.shuf", "-o", "ref_sk", "fa"], cwd=d, timeout=1800)
            t_ref = time.time() - t0
            sets = ko.sketch_sets_by_name(os.path.join(d, "ref_sk"))
            for i in range(n):
                assert np.array_equal(sets["r00_g%04d.fasta" % i], gpu_sets[i]), "reference binary sketch != GPU sketch"
            out["reference"] = {"value": nf / t_ref, "unit": "genomes/s", "cores": p_ref, "kind": "reference",
                                "sample": fa_desc + ", `oracle/_ref/kssd dist -p %d -L L3K10.shuf` wall time incl. process start "
                                          "and the 64 MiB .shuf load" % p_ref,
                                "mbase_per_s": nb * reps / 1e6 / t_ref}
            # stage II and the search of the reference get slower with very many threads (an omp region per query row,
            # command_dist.c:1238; measured on a 256-thread box: 118 s at -p 256): they run with at most 16
            p_srch = max(1, min(p_ref, 16))
            t0 = time.time()
            rs = "ref_sk"
            if sub:
                ko.run_ref(["dist", "-p", max(1, min(cores, ns - 1)), "-L", "L3K10.shuf", "-o", "ref_sk_s", "fa_s"], cwd=d, timeout=1800)
                rs = "ref_sk_s"
                t0 = time.time()
            ko.run_ref(["dist", "-p", p_srch, "-o", "ref_idx", rs], cwd=d, timeout=1800)
            t_idx = time.time() - t0
            t0 = time.time()
            ko.run_ref(["dist", "-p", p_srch, "-r", "ref_idx", "-o", "ref_dist0", rs], cwd=d, timeout=1800)
            t_srch0 = time.time() - t0
            t0 = time.time()
            ko.run_ref(["dist", "-p", p_srch, "-r", "ref_idx", "-o", "ref_dist", "--keepskf", rs], cwd=d, timeout=1800)
            t_srch = min(time.time() - t0, t_srch0)      # the better of two runs, like ours (the second one keeps sharedk_ct.dat for the parity check)
            if ngz:
                t0 = time.time()
                ko.run_ref(["dist", "-p", max(1, min(cores, ngz - 1)), "-L", "L3K10.shuf", "-o", "ref_gz", "gz"], cwd=d, timeout=1800)
                t_gz = time.time() - t0
                out["reference_gz"] = {"value": ngz / t_gz, "unit": "genomes/s", "cores": max(1, min(cores, ngz - 1)), "kind": "reference", "seconds": t_gz,
                                       "sample": "the same %d .fasta.gz files, `oracle/_ref/kssd dist` (zcat -fc pipes)" % ngz}
            out["dist_reference"] = {"value": ns * ns / t_srch, "unit": "pairs/s", "cores": p_srch, "kind": "reference",
                                     "sample": "%d x %d all-pairs of the reference's own sketches of those files: `kssd dist -r <mco> "
                                               "--keepskf <co>` wall time incl. distance.out text; its stage II (2 GiB mco.index, "
                                               "co2mco.c:57-62) took %.1f s on top and is not in the figure" % (ns, ns, t_idx),
                                     "stage2_seconds": t_idx, "seconds": t_srch}
            if "end_to_end" in out:
                # parity of the whole product path at this size: our command line, fed the REFERENCE's sketch directory, must
                # leave the reference's sharedk_ct.dat and distance.out byte for byte
                _run_ours(["dist", "-p", cores, "-r", rs, "-o", "our_dist_on_ref", "--keepskf", rs], d)
                for fn in ("sharedk_ct.dat", "distance.out"):
                    a = open(os.path.join(d, "our_dist_on_ref", fn), "rb").read()
                    b = open(os.path.join(d, "ref_dist", fn), "rb").read()
                    assert a == b, "kssd CLI %s differs from the reference's" % fn
                out["end_to_end"]["search"]["byte_identical_to_reference"] = "sharedk_ct.dat and distance.out on the reference's sketch directory"
                out["end_to_end"]["search"]["speedup_vs_reference"] = t_srch / out["end_to_end"]["search"]["seconds"]
                # ---- the same search where start-up no longer dominates: 4 x the sketches (the directory above under four names each),
                # both binaries on ONE sketch directory, distance.out compared byte for byte (1.9 GB of text at 4 096 x 4 096)
                n4 = e2e_search4k if ns >= 256 else 0
                if n4 >= 2 * ns:
                    import filecmp
                    for junk in ("ref_idx", "ref_dist0", "ref_dist", "our_dist_on_ref", "our_dist0", "our_dist", "our_dist_k"):
                        shutil.rmtree(os.path.join(d, junk), ignore_errors=True)      # (the 2 GiB mco.index and the first leg's reports)
                    S = K.SketchSet.read(os.path.join(d, rs))
                    times = n4 // ns
                    sizes = np.diff(S.off)
                    big = K.SketchSet(S.shuf_id, S.kmerlen, S.dim_rd_len, S.comp_num,
                                      ["x%02d/%s" % (r, os.path.basename(nm)) for r in range(times) for nm in S.names],
                                      np.concatenate([[0], np.cumsum(np.tile(sizes, times))]).astype(np.uint64), np.tile(S.ids, times))
                    big.write(os.path.join(d, "sk4k"), K.derive(10, 6, 3).hashsize, slot_order=False)
                    nq = times * ns
                    dt4a, _ = _run_ours(["dist", "-p", cores, "-r", "sk4k", "-o", "our_d4k0", "sk4k"], d)
                    shutil.rmtree(os.path.join(d, "our_d4k0"), ignore_errors=True)
                    dt4, tm4 = _run_ours(["dist", "-p", cores, "-r", "sk4k", "-o", "our_d4k", "sk4k"], d, {"KSSD_TIMING": "1"})
                    t0 = time.time()
                    ko.run_ref(["dist", "-p", p_srch, "-o", "ref_idx4k", "sk4k"], cwd=d, timeout=1800)
                    t_idx4 = time.time() - t0
                    t0 = time.time()
                    ko.run_ref(["dist", "-p", p_srch, "-r", "ref_idx4k", "-o", "ref_d4k", "sk4k"], cwd=d, timeout=1800)
                    t_ref4 = time.time() - t0
                    same = filecmp.cmp(os.path.join(d, "our_d4k", "distance.out"), os.path.join(d, "ref_d4k", "distance.out"), shallow=False)
                    assert same, "kssd CLI distance.out differs from the reference's at %d x %d" % (nq, nq)
                    mb = os.path.getsize(os.path.join(d, "our_d4k", "distance.out")) >> 20
                    out["end_to_end"]["search_4096"] = {
                        "value": nq * nq / min(dt4, dt4a), "unit": "pairs/s", "seconds": min(dt4, dt4a), "seconds_runs": [dt4a, dt4], "stages": tm4,
                        "reference_seconds": t_ref4, "reference_stage2_seconds": t_idx4, "reference_cores": p_srch,
                        "speedup_vs_reference": t_ref4 / min(dt4, dt4a),
                        "byte_identical_to_reference": "distance.out (%d MB)" % mb,
                        "what": "`kssd dist -r <sketches> -o <dir> <sketches>` at %d x %d (the %d sketches above under %d names each): wall time of the "
                                "command incl. process start and the %d MB of distance.out, the better of two runs; the reference binary on the same "
                                "directory, one run, its stage II (%.1f s) not counted" % (nq, nq, ns, times, mb, t_idx4)}
    finally:
        shutil.rmtree(d, ignore_errors=True)
    return out


# ------------------------------------------------------------------------------------------------------
# BASELINE configs[3]: reads -> read-set sketch (fastq2co, iseq2comem.c:277-356) -> containment (-M 1) against the
# reference sketches.  The reads are generated and packed on the device exactly as the host tokeniser lays a FASTQ
# file out (kssd_batch_add_fastq: the reads of a file are ONE genome, one invalid position between two reads).
# ------------------------------------------------------------------------------------------------------




def fastq_end_to_end(shuf, fq, n_reads, sk, ko):
    """the slice as ONE .fastq file in tmpfs through the product's command line and through the reference binary: wall
    times, and combco.0 byte for byte (the oracle's file order / the reference's file)"""
    out = {}
    d = tempfile.mkdtemp(prefix="kssd_benchq_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        with open(os.path.join(d, "reads.fastq"), "wb") as f:
            f.write(fq)
        shuf.write(os.path.join(d, "L3K10.shuf"))
        gbase = n_reads * READ_LEN / 1e9
        desc = "%d reads x %d bp as one %.2f GB .fastq file in tmpfs" % (n_reads, READ_LEN, len(fq) / 1e9)
        want = None
        if os.access(KSSD_BIN, os.X_OK):
            dt_first, tm_first = _run_ours(["dist", "-p", host_cores(), "-L", "L3K10.shuf", "-o", "our_sk0", "reads.fastq"], d, {"KSSD_TIMING": "1"})
            dt, tm = _run_ours(["dist", "-p", host_cores(), "-L", "L3K10.shuf", "-o", "our_sk", "reads.fastq"], d, {"KSSD_TIMING": "1"})
            runs = [dt_first, dt]
            if dt_first < dt:
                dt, tm = dt_first, tm_first
            want = sk.fastq(fq, Q=0, M=1)     # the reference's file order
            got = np.fromfile(os.path.join(d, "our_sk", "combco.0"), np.uint32)
            assert np.array_equal(got, want), "kssd CLI combco.0 of the read set != oracle (file order)"
            out["end_to_end"] = {"value": gbase / dt, "unit": "Gbase/s", "seconds": dt, "reads_per_s": n_reads / dt, "stages": tm,
                                 "what": "`kssd dist -L L3K10.shuf -o <dir> reads.fastq`: process start, .shuf load, file read into a "
                                         "page-locked buffer, raw text H2D, tokenised + sketched on the device, D2H, slot order, combco.* "
                                         "written -- wall time of the command, the better of two runs (HIP start-up varies by ~0.1 s from "
                                         "process to process); combco.0 equals the oracle's ids in file order",
                                 "sample": desc, "seconds_runs": runs}
            dt_h, _ = _run_ours(["dist", "-p", host_cores(), "-L", "L3K10.shuf", "-o", "our_sk_host", "reads.fastq"], d, {"KSSD_HOST_FASTQ": "1"})
            out["end_to_end"]["seconds_with_host_tokeniser"] = dt_h
        if ko.have_ref() and shutil.which("zcat"):
            t0 = time.time()
            ko.run_ref(["dist", "-L", "L3K10.shuf", "-o", "ref_sk", "reads.fastq"], cwd=d, timeout=1800)
            t_ref = time.time() - t0
            ref = np.fromfile(os.path.join(d, "ref_sk", "combco.0"), np.uint32)
            if want is not None:
                assert np.array_equal(ref, want), "reference binary combco.0 != oracle"
            out["cpu_baseline_reference"] = {"value": gbase / t_ref, "unit": "Gbase/s", "cores": 1, "kind": "reference", "seconds": t_ref,
                                             "sample": desc + ", `oracle/_ref/kssd dist -L L3K10.shuf` wall time (one FASTQ file = one thread, "
                                                              "command_dist.c:275)"}
            if "end_to_end" in out:
                out["end_to_end"]["byte_identical_to_reference"] = "combco.0"
    finally:
        shutil.rmtree(d, ignore_errors=True)
    return out


def run_fastq(a, shuf, dev):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    G, L, NSRC = a.genomes, a.length, 64
    t0 = time.time()
    packed, mask, chunk_off, kept = make_batch(G, L, a.clades, 20260101, dev, keep_codes=min(NSRC, G), keep_on_device=True)
    ctx = K.GpuCtx(shuf, dev.index or 0)
    cap = int(G * L / 4096 * 1.25) + 4096
    roff = torch.zeros(G + 1, dtype=torch.int64, device=dev)
    rids = torch.zeros(cap, dtype=torch.int32, device=dev)
    for attempt in range(6):
        ctx.sketch_device(packed, mask, chunk_off, roff, rids, cap)
        rc, rtotal, bad = ctx.sketch_status()
        if rc == 0:
            break
    assert rc == 0, rc
    del packed, mask
    torch.cuda.empty_cache()
    src = [c for c, _ in kept]
    log("[bench] %d reference genomes sketched (%d ids) in %.1f s" % (G, rtotal, time.time() - t0))
    t0 = time.time()
    n_par = min(a.parity_reads, a.reads)
    rp, rm, rco, host_reads = make_reads_batch(src, a.reads, 4242, dev, keep_reads=n_par)
    n_bases = a.reads * READ_LEN
    n_pos = a.reads * (READ_LEN + 1)
    torch.cuda.synchronize()
    log("[bench] %d reads packed on device in %.1f s (%d chunks)" % (a.reads, time.time() - t0, int(rco[1])))
    # the reference index (untimed setup; `kssd dist -r` finds it prebuilt as mco.* too)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    # (the query is ONE row of a read set's ids, most of which are sequencing errors and miss the index: the negative filter in front of
    # the table, as the host-level searches switch it on for such query sets -- csrc/kssd_dist.inc: dist_queries_miss_heavy)
    ctx.index_set_filter(True, 0, 0)
    ctx.index_build_device(roff, rids, G, int(rtotal))
    e1.record()
    torch.cuda.synchronize()
    index_ms = e0.elapsed_time(e1)
    qcap = int(n_pos / 4096 * 1.5) + 4096
    rsumm = None if a.no_mask_summary else mask_summary(ctx, rm, int(rco[-1]), dev)
    res_m = {}
    for M in (1, 2):
        qoff = torch.zeros(2, dtype=torch.int64, device=dev)
        qids = torch.zeros(qcap, dtype=torch.int32, device=dev)
        shared = torch.zeros(G, dtype=torch.int32, device=dev)
        cont = torch.zeros(G, dtype=torch.float64, device=dev)
        aaf = torch.zeros(G, dtype=torch.float64, device=dev)
        flags = K.SKETCH_KEEP_ZERO | K.SKETCH_NO_CAPACITY

        def step(timed=None):
            ctx.sketch_plan(rp, rm, rco, qoff, qids, qcap, flags, M, d_summary=rsumm)
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)] if timed is not None else None
            for i, ph in enumerate((K.PHASE_PREP, K.PHASE_SCAN, K.PHASE_EXACT, K.PHASE_FINISH)):
                if ev:
                    ev[i].record()
                ctx.sketch_phase(ph, None)
            if ev:
                ev[4].record()
            ctx.dist_device(qoff, qids, 1, 0, 1, shared, None, None, cont, aaf, max_row_ids=qcap)
            if ev:
                ev[5].record()
                timed.append(ev)
        for attempt in range(8):                          # sizes the workspaces
            step()
            rc, qtotal, bad = ctx.sketch_status()
            if rc == 0:
                break
            assert rc == K.capi.ERR_OVERFLOW, rc
        assert rc == 0
        for _ in range(a.warmup):
            step()
        torch.cuda.synchronize()
        ctx.kernel_time(0, reset=True)
        timed = []
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step(timed)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        scan_ms, scan_n = ctx.kernel_time(0)
        ph = np.array([[ev[i].elapsed_time(ev[i + 1]) for i in range(5)] for ev in timed]).mean(0)
        rc, qtotal, bad = ctx.sketch_status()
        assert rc == 0
        res_m[M] = dict(dt=dt, scan_ms=scan_ms, phases=ph, qtotal=int(qtotal), qids=qids[:int(qtotal)].cpu().numpy().view(np.uint32),
                        shared=shared.cpu().numpy().view(np.uint32), cont=cont.cpu().numpy(), aaf=aaf.cpu().numpy())
    # ---- parity: the oracle on a slice of the same reads (FASTQ text through the host tokeniser), CPU baseline beside it
    import kssd_oracle as ko
    par = None
    cpu = None
    if n_par:
        from synth import fastq_records
        t0 = time.time()
        fq = fastq_records(host_reads)
        hb = K.Batch()
        t1 = time.time()
        assert hb.add_fastq(fq, Q=0) == 4 * n_par
        t_host_tok = time.time() - t1
        # the device-side generator writes what the tokeniser writes: same packed words and mask for the slice
        nchk = hb.n_chunks
        whole = (n_par * (READ_LEN + 1)) // K.CHUNK_BASES    # chunks that hold slice reads only
        assert np.array_equal(hb.packed()[:whole * K.CHUNK_WORDS], rp[:whole * K.CHUNK_WORDS].cpu().numpy().view(np.uint32))
        assert np.array_equal(hb.mask()[:whole * K.CHUNK_MASKW], rm[:whole * K.CHUNK_MASKW].cpu().numpy().view(np.uint32))
        t_tok = time.time() - t0
        sk = ko.Sketcher(shuf.table, 10, 6, 3)
        par = {"reads": n_par, "mbase": n_par * READ_LEN / 1e6}
        for M in (1, 2):
            t0 = time.time()
            want = np.sort(sk.fastq(fq, Q=0, M=M))
            t_or = time.time() - t0
            off_s, ids_s = ctx.sketch_batch(hb, K.SKETCH_KEEP_ZERO | K.SKETCH_NO_CAPACITY, min_occ=M)
            assert np.array_equal(ids_s, want), "fastq slice -n %d: GPU sketch != oracle (%d vs %d ids)" % (M, len(ids_s), len(want))
            par["n%d_ids" % M] = int(len(want))
            if M == 1:
                cpu = {"value": n_par * READ_LEN / 1e9 / t_or, "unit": "Gbase/s", "cores": 1, "kind": "port",
                       "sample": "%d of the reads (%.0f Mbase) as FASTQ text in memory, oracle/kssd_oracle.c ko_fastq2co on one "
                                 "thread (the reference sketches one FASTQ file on one thread, command_dist.c:275)" % (n_par, n_par * READ_LEN / 1e6)}
        # ---- the same slice as FASTQ TEXT in HBM through the device tokeniser (csrc/kssd_tok.inc): what the command line does
        d_text = torch.from_numpy(np.frombuffer(fq, dtype=np.uint8).copy()).to(dev)
        tco = np.array([0, (len(fq) + K.CHUNK_BASES - 1) // K.CHUNK_BASES], dtype=np.uint64)
        tp = torch.zeros(int(tco[1]) * K.CHUNK_WORDS + K.SLACK_WORDS, dtype=torch.int32, device=dev)
        tm = torch.zeros(int(tco[1]) * K.CHUNK_MASKW + K.SLACK_WORDS, dtype=torch.int32, device=dev)
        tok_ms = []
        for it in range(4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            rc, bad, npos_t, nlines_t = ctx.tokenise_fastq_device(d_text, [0], [len(fq)], tp, tm, tco)   # synchronises
            tok_ms.append((time.perf_counter() - t0) * 1e3)
            assert rc == 0 and int(nlines_t[0]) == 4 * n_par and int(npos_t[0]) == hb.n_positions(0)
        nw, nm = nchk * K.CHUNK_WORDS, nchk * K.CHUNK_MASKW
        assert np.array_equal(tp[:nw].cpu().numpy().view(np.uint32), hb.packed()[:nw]), "device FASTQ tokeniser: packed words != host tokeniser"
        assert np.array_equal(tm[:nm].cpu().numpy().view(np.uint32), hb.mask()[:nm]), "device FASTQ tokeniser: mask words != host tokeniser"
        par["device_tokeniser"] = {"text_bytes": len(fq), "ms": min(tok_ms[1:]), "gb_per_s": len(fq) / 1e9 / (min(tok_ms[1:]) * 1e-3),
                                   "host_tokeniser_s_one_thread": t_host_tok,
                                   "what": "FASTQ text of the slice resident in HBM -> packed batch (csrc/kssd_tok.inc, 9 launches), wall time of the call incl. its "
                                           "status read-back; output bit-identical to libkssd_host.so's tokeniser"}
        del d_text, tp, tm
        hb.close()
        e2e = fastq_end_to_end(shuf, fq, n_par, sk, ko)
        del fq
    # containment rows of the full run against the oracle's posting traversal
    oh = roff.cpu().numpy().astype(np.uint64)
    ih = rids[:int(rtotal)].cpu().numpy().view(np.uint32)
    szh = np.diff(oh).astype(np.uint32)
    for M in (1, 2):
        r = res_m[M]
        qo = np.array([0, r["qtotal"]], dtype=np.uint64)
        want = ko.shared_counts(oh, ih, qo, r["qids"], threads=host_cores())
        assert np.array_equal(r["shared"][None, :], want), "containment row -n %d: shared counts != oracle" % M
        oJ, oMD, oC, oAD = ko.metrics_batch(szh[None, :], np.array([[r["qtotal"]]], dtype=np.uint32), want, 20)
        assert np.array_equal(r["cont"][None, :].view(np.int64), oC.view(np.int64))
        assert np.abs(r["aaf"][None, :].view(np.int64) - oAD.view(np.int64)).max() <= 1
    r1 = res_m[1]
    scan_bytes = 0.375 * n_pos + 4.0 * r1["qtotal"]
    achieved = scan_bytes / (r1["scan_ms"] * 1e-3) / 1e9
    res = {
        "metric": "Gbase of reads sketched/s (fastq2co sketch of one read set + containment row against the reference sketches, L3K10)",
        "value": n_bases * a.steps / r1["dt"] / 1e9, "unit": "Gbase/s",
        "n_gpus": 1, "steps": a.steps, "warmup": a.warmup, "ms_per_step": r1["dt"] / a.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": "BASELINE configs[3]: %d x %d bp reads from %d of %d reference genomes (%.1f Mb, %d clades), 0.5 %% "
                               "substitutions, both strands, -n 1; containment (-M 1) against the %d reference sketches"
                               % (a.reads, READ_LEN, len(src), G, L / 1e6, a.clades, G),
                   "k": 10, "subk": 6, "drlevel": 3, "reads": a.reads, "read_len": READ_LEN, "references": G},
        "reads_per_s": a.reads * a.steps / r1["dt"],
        "n1": {"ids": r1["qtotal"], "ms_per_step": r1["dt"] / a.steps * 1e3,
               "phase_ms": dict(zip(["prep", "scan", "exact", "dedup_finish(rocPRIM sort path)", "containment_row"], [float(x) for x in r1["phases"]]))},
        "n2": {"ids": res_m[2]["qtotal"], "ms_per_step": res_m[2]["dt"] / a.steps * 1e3, "gbase_per_s": n_bases * a.steps / res_m[2]["dt"] / 1e9,
               "phase_ms": dict(zip(["prep", "scan", "exact", "dedup_finish(rocPRIM sort path)", "containment_row"], [float(x) for x in res_m[2]["phases"]]))},
        "reference_index_build_ms": index_ms, "reference_ids": int(rtotal),
        "best_hit_shared": int(r1["shared"].max()), "source_genomes_min_shared": int(r1["shared"][:len(src)].min()),
        "roofline": {"bound": "hbm", "kernel": "sketch_scan_kernel<6>", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes_per_launch": scan_bytes},
        "parity": par,
    }
    if cpu:
        res["cpu_baseline"] = cpu
    if n_par and e2e:
        res.update(e2e)
    print(json.dumps(res), flush=True)
    ctx.close()


# ------------------------------------------------------------------------------------------------------
# BASELINE configs[4]: 3 Gb "mammalian" records at the s7/l5 shuffle (`kssd shuffle -k 10 -s 7 -l 5`: SURVEY.md section 8 --
# auto -L 5 is rejected by the reference, and its effective reduction is 16^4).  Sketch throughput only (the config's
# metric); no exchange between ranks: N ranks are N replicas with their own records.
# The reference itself cannot sketch such a record: hashlimit 4 914 < the ~45 800 ids of 3 Gb, it aborts
# (iseq2comem.c:262-263).  So the whole-record run lifts the capacity rule (KSSD_SKETCH_NO_CAPACITY) and parity is
# (a) record 0 cut into <= 250 Mb pieces (overlapping by 2k - 1 bases: no k-mer lost, none invented) through the oracle
#     (and the reference binary when the snapshot carries it): the UNION of their id sets must be record 0's sketch;
# (b) the first 400 Mb of record 0 as a record of its own WITHOUT the flag: the reference's abort, naming the genome.
# ------------------------------------------------------------------------------------------------------


# ------------------------------------------------------------------------------------------------------
# roofline_tok: what the product runs on the device in FRONT of the scan.  `kssd dist` uploads raw FASTA text and tokenises it on the
# device (csrc/kssd_tok.inc: the byte rules of fasta2co, iseq2comem.c:213-242); the headline's batch is the packed form.  Here the
# same genomes as resident FASTA text (header line, 70 columns, N's) go through kssd_gpu_tokenise_fasta_device, its two passes over
# the text bracketed by their own HIP events like the scan, and the tokenised batch is sketched: the CSR must be the headline's.
# Algorithmic bytes: 1.0 B per text byte in + 0.375 B per position out.
# ------------------------------------------------------------------------------------------------------


def tokeniser_leg(a, shuf, local, dev, sink, m):
    G = len(sink.len)
    ctx = K.GpuCtx(shuf, local)
    tstream = torch.cuda.current_stream()
    stream = tstream.cuda_stream
    chunks = (sink.len + np.uint64(4095)) // np.uint64(4096)
    tco = np.concatenate([[0], np.cumsum(chunks)]).astype(np.uint64)
    n_chunks = int(tco[-1])
    tp = torch.zeros(n_chunks * K.CHUNK_WORDS + 64, dtype=torch.int32, device=dev)
    tm = torch.zeros(n_chunks * K.CHUNK_MASKW + 64, dtype=torch.int32, device=dev)
    tsumm = torch.zeros(max(n_chunks, 1), dtype=torch.int64, device=dev)   # the mask's summary words, written by the tokeniser with the mask
    rc, bad, npos = ctx.tokenise_fasta_device(sink.text, sink.off, sink.len, tp, tm, tco, stream, d_summary=tsumm)      # (synchronises; also the warm-up)
    if rc != 0:
        raise SystemExit("tokeniser leg: rc=%d file %d" % (rc, bad))
    text_bytes, positions = int(sink.len.sum()), int(npos.sum())
    reps = max(3, min(10, a.steps))
    ctx.tokenise_fasta_device(sink.text, sink.off, sink.len, tp, tm, tco, stream, status=False, d_summary=tsumm)
    torch.cuda.synchronize()
    ctx.kernel_time(2, reset=True)
    ctx.kernel_time(3, reset=True)
    ctx.set_kernel_timing(1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(tstream)
    for _ in range(reps):
        ctx.tokenise_fasta_device(sink.text, sink.off, sink.len, tp, tm, tco, stream, status=False, d_summary=tsumm)
    e1.record(tstream)
    torch.cuda.synchronize()
    call_ms = e0.elapsed_time(e1) / reps
    t_sum, t_emit = ctx.kernel_times(2), ctx.kernel_times(3)
    ctx.set_kernel_timing(0)
    # the tokenised batch's sketches are the headline's (same genomes; the packed words differ where two N's meet: one invalid position)
    cap = len(m["ids"])
    off = torch.zeros(G + 1, dtype=torch.int64, device=dev)
    ids = torch.zeros(cap, dtype=torch.int32, device=dev)
    for attempt in range(6):
        ctx.sketch_device(tp, tm, tco, off, ids, cap, d_summary=tsumm)     # (scanned with the tokeniser's own summary words: the product's path)
        rc, total, _ = ctx.sketch_status()
        if rc == 0:
            break
        assert rc == K.capi.ERR_OVERFLOW, rc
    assert rc == 0 and int(total) == m["total"], (rc, total, m["total"])
    assert np.array_equal(off.cpu().numpy().view(np.uint64), np.asarray(m["off"]).view(np.uint64)), "tokeniser leg: sketch sizes differ from the packed batch's"
    assert np.array_equal(ids.cpu().numpy().view(np.uint32)[:int(total)], m["ids"][:int(total)]), "tokeniser leg: ids differ from the packed batch's"
    # how much of the mask the tokeniser's summary words spare the scan: the lanes it leaves clear against the exact count
    exact = torch.zeros(max(n_chunks, 1), dtype=torch.int64, device=dev)
    ctx.mask_summarise_device(tm, n_chunks, exact)
    torch.cuda.synchronize()
    assert bool(((tsumm & ~exact) == 0).all()), "tokeniser leg: a summary bit is set where the mask holds a run-breaking position"
    lanes_tok, lanes_exact = summary_clear_lanes(tsumm, n_chunks), summary_clear_lanes(exact, n_chunks)
    ctx.close()
    del tp, tm
    torch.cuda.empty_cache()
    one_pass = len(t_emit) == 0      # (round 6: FASTA goes through tok_onepass_kernel; KSSD_TOK_TWO_PASS=1 keeps the two passes of rounds 2 - 5)
    kern_ms = float(t_sum.mean()) if one_pass else float(t_sum.mean() + t_emit.mean())
    alg = 1.0 * text_bytes + 0.375 * positions
    ach = alg / (kern_ms * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": "tok_onepass_kernel" if one_pass else "tok_summarise_kernel<false> + tok_emit_kernel<false>", "achieved": ach,
            "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": ach / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes_per_launch": alg,
            "bytes_moved": (1.0 if one_pass else 2.0) * text_bytes + 0.375 * n_chunks * 4096 * 2,
            "kernel_ms": kern_ms, "min_ms": float(t_sum.min()) if one_pass else None, "max_ms": float(t_sum.max()) if one_pass else None,
            "summarise_ms": None if one_pass else float(t_sum.mean()), "emit_ms": None if one_pass else float(t_emit.mean()),
            "launches_timed": [len(t_sum), len(t_emit)],
            "call_ms": call_ms, "text_bytes": text_bytes, "positions": positions,
            "summary_lanes_clear": {"tokeniser": lanes_tok, "exact": lanes_exact, "lanes": n_chunks * 64,
                                    "what": "lanes of 64 positions whose summary bit the tokeniser leaves clear (the scan fetches their mask words) "
                                            "against the lanes that really hold a run-breaking position or padding"},
            "text_GBs_whole_call": text_bytes / (call_ms * 1e-3) / 1e9,
            "what": "the device ingests ASCII here (1.0 B per text byte + 0.375 B per position written), the headline's scan ingests the packed form "
                    "(0.375 B/base); `kssd dist` runs this in front of every scan (a-3: iseq2comem.c:213-242).  achieved = algorithmic bytes / the "
                    "tokenising kernel's launch duration (events of the dispatch itself; two passes: their sum); call_ms = the whole call incl. the "
                    "zeroing of the outputs; bytes_moved: the text read (twice in two passes), the outputs zeroed and written.  "
                    "The tokenised batch's sketches equal the packed batch's (checked)."}


def run_mammal(a, dev, world, rank):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    G, L = a.genomes, a.length
    t0 = time.time()
    shuf = K.Shuf.generate(10, 7, 5, seed=20260105)
    info = K.derive(10, 7, 5)
    assert info.hashsize == 8191 and info.hashlimit == 4914       # primer[5], 0.6 of it
    do_par = rank == 0 and a.cpu_sample > 0
    packed, mask, chunk_off, (c0, n0) = make_long_records(G, L, 20260105 + 7919 * rank, dev, keep_first=do_par)
    torch.cuda.synchronize()
    if rank == 0:
        log("[bench] %d records x %.2f Gb packed on device in %.1f s (%.1f GB of packed bases + mask)"
            % (G, L / 1e9, time.time() - t0, (packed.numel() + mask.numel()) * 4 / 1e9))
    ctx = K.GpuCtx(shuf, dev.index or 0)
    n_pos = G * L
    cap = int(n_pos / 65536 * 1.3) + 65536
    off_d = torch.zeros(G + 1, dtype=torch.int64, device=dev)
    ids_d = torch.zeros(cap, dtype=torch.int32, device=dev)
    flags = K.SKETCH_FASTA | K.SKETCH_NO_CAPACITY
    summ = None if a.no_mask_summary else mask_summary(ctx, mask, int(chunk_off[-1]), dev)

    def step():
        ctx.sketch_plan(packed, mask, chunk_off, off_d, ids_d, cap, flags, 1, d_summary=summ)
        for ph in (K.PHASE_PREP, K.PHASE_SCAN, K.PHASE_EXACT, K.PHASE_FINISH):
            ctx.sketch_phase(ph, None)
    for attempt in range(8):                                       # sizes the workspaces
        step()
        rc, total, bad = ctx.sketch_status()
        if rc == 0:
            break
        assert rc == K.capi.ERR_OVERFLOW, rc
    assert rc == 0
    if world > 1:
        import torch.distributed as dist

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    for _ in range(max(a.warmup, 1)):
        step()
    sync()
    ctx.kernel_time(0, reset=True)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
    t0 = time.perf_counter()
    for n in range(a.steps):
        if n == a.steps - 1:                                       # phase split of the last step (events on the stream the phases run on)
            ctx.sketch_plan(packed, mask, chunk_off, off_d, ids_d, cap, flags, 1, d_summary=summ)
            for i, ph in enumerate((K.PHASE_PREP, K.PHASE_SCAN, K.PHASE_EXACT, K.PHASE_FINISH)):
                ev[i].record()
                ctx.sketch_phase(ph, None)
            ev[4].record()
        else:
            step()
    sync()
    dt = time.perf_counter() - t0
    rc, total, bad = ctx.sketch_status()
    assert rc == 0
    scan_ms, scan_n = ctx.kernel_time(0)
    phases = [ev[i].elapsed_time(ev[i + 1]) for i in range(4)]
    n_stage1, n_bloom = ctx.scan_stats()
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    oh = off_d.cpu().numpy()
    ih = ids_d[:int(total)].cpu().numpy().view(np.uint32)
    # size-independent properties on every record: ascending distinct ids, the expected sampling rate (2^-16 per position)
    for gi in range(G):
        x = ih[int(oh[gi]):int(oh[gi + 1])].astype(np.int64)
        assert np.all(np.diff(x) > 0), "record %d: ids not ascending and distinct" % gi
        assert abs(len(x) / (L / 65536.0) - 1.0) < 0.05, "record %d: %d ids for %d positions" % (gi, len(x), L)
    res = None
    if rank == 0:
        scan_bytes = 0.375 * n_pos + 4.0 * total
        achieved = scan_bytes / (scan_ms * 1e-3) / 1e9
        res = {
            "metric": "Gbase sketched/s (fasta2co sketch of 3 Gb records at -k 10 -s 7 -l 5, whole records, capacity rule lifted)",
            "value": world * n_pos * a.steps / dt / 1e9, "unit": "Gbase/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": "BASELINE configs[4]: %d synthetic %.2f Gb records per GPU (uniform bases, 1e-5 N), shuffle -k 10 -s 7 -l 5 "
                                   "(the runnable form of 'L5K10': SURVEY.md section 8), sketch only, KSSD_SKETCH_NO_CAPACITY" % (G, L / 1e9),
                       "k": 10, "subk": 7, "drlevel": 5, "records_per_gpu": G, "record_len": L,
                       "parallelism": "single GPU" if world == 1 else "%d replicas, records sharded, no collective" % world},
            "genomes_per_s": world * G * a.steps / dt, "ids_per_batch": int(total), "ids_per_record": int(total) / G,
            "phase_ms_last_step": dict(zip(["prep", "scan", "exact", "dedup_finish (LDS sort in parts, offsets, gather)"], [float(x) for x in phases])),
            "kernels": {"sketch_scan_ms": scan_ms, "launches_timed": scan_n,
                        "scan_positions_past_stage1": n_stage1 / n_pos, "scan_positions_past_bloom": n_bloom / n_pos},
            "roofline": {"bound": "hbm", "kernel": "sketch_scan_kernel<7>", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes_per_launch": scan_bytes},
        }
    if do_par:
        import kssd_oracle as ko
        from synth import fasta_text
        cores = host_cores()
        got0 = ih[int(oh[0]):int(oh[1])]
        # (b) the reference's capacity abort on a 400 Mb record (its own genome index in the error)
        c400 = (400_000_000 // K.CHUNK_BASES)
        co2 = np.array([0, 256, 256 + c400], dtype=np.uint64)        # a small record in front: the abort must name record 1
        o2 = torch.zeros(3, dtype=torch.int64, device=dev)
        i2 = torch.zeros(65536, dtype=torch.int32, device=dev)
        rc2 = None
        for attempt in range(8):
            ctx.sketch_device(packed, mask, co2, o2, i2, 65536)
            rc2, tot2, bad2 = ctx.sketch_status()
            if rc2 != K.capi.ERR_OVERFLOW:
                break
        assert rc2 == K.capi.ERR_CAPACITY and bad2 == 1, "400 Mb record: expected the reference's capacity abort, got rc %r genome %r" % (rc2, bad2)
        # (a) record 0 in <= 250 Mb pieces through the oracle
        P = 250_000_000
        ov = 2 * 10 - 1
        t0 = time.time()
        texts = []
        for s0 in range(0, L, P):
            b = max(0, s0 - ov)
            texts.append(fasta_text(c0[b:s0 + P], b"rec0_%d" % (s0 // P), n_mask=n0[b:s0 + P]))
        t_text = time.time() - t0
        t0 = time.time()
        thr = min(cores, len(texts))
        ooff, oids = ko.sketch_texts(shuf.table, 10, 7, 5, texts, threads=thr)
        t_or = time.time() - t0
        union = np.unique(oids)
        assert np.array_equal(got0, union), "record 0: GPU sketch (%d ids) != union of the oracle's pieces (%d ids)" % (len(got0), len(union))
        res["parity"] = {"record0_ids": int(len(got0)), "pieces": len(texts), "piece_len": P, "oracle_seconds": t_or,
                         "capacity_abort_400Mb": "KSSD_ERR_CAPACITY naming record 1 (hashlimit 4914), as iseq2comem.c:262-263",
                         "what": "record 0's sketch == union of the oracle's id sets of its %d overlapping pieces (bit-exact)" % len(texts)}
        res["cpu_baseline"] = {"value": L / 1e9 / t_or, "unit": "Gbase/s", "cores": thr, "kind": "port",
                               "sample": "record 0 (%.2f Gbase) as %d FASTA texts of <= 250 Mb in memory, oracle/kssd_oracle.c "
                                         "sketch_texts, OpenMP over the pieces (the 1 GiB table of -s 7 misses every cache)" % (L / 1e9, len(texts))}
        # the same pieces as files: the product's command line and, when the snapshot carries it, the reference binary
        d = tempfile.mkdtemp(prefix="kssd_benchm_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
        try:
            os.mkdir(os.path.join(d, "fa"))
            for i, t in enumerate(texts):
                with open(os.path.join(d, "fa", "rec0_%02d.fasta" % i), "wb") as f:
                    f.write(t)
            nb_txt = sum(len(t) for t in texts)
            n_pieces = len(texts)
            del texts
            shuf.write(os.path.join(d, "s7l5.shuf"))
            if os.access(KSSD_BIN, os.X_OK):
                dt1, tm1 = _run_ours(["dist", "-p", cores, "-L", "s7l5.shuf", "-o", "our_sk0", "fa"], d, {"KSSD_TIMING": "1"})
                dt2, tm2 = _run_ours(["dist", "-p", cores, "-L", "s7l5.shuf", "-o", "our_sk", "fa"], d, {"KSSD_TIMING": "1"})
                runs = [dt1, dt2]
                if dt1 < dt2:
                    dt2, tm2 = dt1, tm1
                ours = ko.sketch_sets_by_name(os.path.join(d, "our_sk"))
                u = np.unique(np.concatenate([ours["rec0_%02d.fasta" % i] for i in range(n_pieces)]))
                assert np.array_equal(u, got0), "kssd CLI: union of the pieces' sketches != record 0's sketch"
                for i in range(n_pieces):
                    assert np.array_equal(np.sort(ours["rec0_%02d.fasta" % i]), np.sort(oids[int(ooff[i]):int(ooff[i + 1])])), "kssd CLI piece %d != oracle" % i
                res["end_to_end"] = {"value": L / 1e9 / dt2, "unit": "Gbase/s", "seconds": dt2, "seconds_runs": runs, "stages": tm2,
                                     "what": "`kssd dist -L s7l5.shuf -o <dir> <dir of %d FASTA files, %.2f GB>`: wall time of the command incl. the "
                                             "1 GiB .shuf (second run: its 16 KiB core), the better of two runs; every piece's sketch equals the oracle's"
                                             % (n_pieces, nb_txt / 1e9)}
            if ko.have_ref() and shutil.which("zcat"):
                p_ref = max(1, min(cores, n_pieces - 1))            # the reference only goes parallel with more files than threads (command_dist.c:275)
                t0 = time.time()
                ko.run_ref(["dist", "-p", p_ref, "-L", "s7l5.shuf", "-o", "ref_sk", "fa"], cwd=d, timeout=3000)
                t_ref = time.time() - t0
                sets = ko.sketch_sets_by_name(os.path.join(d, "ref_sk"))
                u = np.unique(np.concatenate(list(sets.values())))
                assert np.array_equal(u, got0), "reference binary: union of the pieces' sketches != record 0's GPU sketch"
                res["cpu_baseline_port"] = res["cpu_baseline"]
                res["cpu_baseline"] = {"value": L / 1e9 / t_ref, "unit": "Gbase/s", "cores": p_ref, "kind": "reference", "seconds": t_ref,
                                       "sample": "record 0 as %d FASTA files of <= 250 Mb in tmpfs, `oracle/_ref/kssd dist -p %d -L s7l5.shuf` wall time "
                                                 "incl. process start and the 1 GiB .shuf load; the whole 3 Gb record it cannot sketch (capacity abort)"
                                                 % (len(sets), p_ref)}
        finally:
            shutil.rmtree(d, ignore_errors=True)
    if rank == 0:
        print(json.dumps(res), flush=True)
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


# ------------------------------------------------------------------------------------------------------


