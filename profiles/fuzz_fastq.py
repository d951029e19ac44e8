"""development tool (GPU box): read sets through kssd_gpu_sketch_batch against the oracle -- fastq2co's rules (-n 1..3) and the abundance
mode (occurrence counts) on random read sets: coverage from 0.2 x to 60 x, reads of 20 .. 300 bases, a hot read thousands of times, Ns,
both strands, several read sets per batch, with the LDS sort and with the large-genome paths forced (kssd_gpu_set_lds_sort_limit).
python3 profiles/fuzz_fastq.py [seeds] [seed base]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in ("", "tests", "oracle"):
    sys.path.insert(0, os.path.join(R, d))
import numpy as np
import kssd_oracle as ko
import public_kssd_amd as K
from synth import fastq_text
PARAMS = [(10, 6, 3), (8, 5, 2), (9, 6, 3)]
n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
base = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = n_cases = 0
for seed in range(n_seeds):
    rng = np.random.default_rng(515_000 + base + seed)
    k, s, l = PARAMS[int(rng.integers(0, len(PARAMS)))]
    shuf = K.Shuf.generate(k, s, l, seed=300 + seed % 7)
    sk = ko.Sketcher(shuf.table, k, s, l)
    ctx = K.GpuCtx(shuf, 0)
    if rng.random() < 0.4:
        ctx.set_lds_sort_limit(int(rng.choice([64, 1024, 4096])))
    sets = []
    for g in range(int(rng.integers(1, 4))):
        G = int(rng.choice([2000, 20_000, 150_000]))
        genome = rng.integers(0, 4, G, dtype=np.uint8)
        cov = float(rng.choice([0.2, 2, 10, 60]))
        L = int(rng.choice([20, 50, 100, 150, 300]))
        L = min(L, G - 1)
        n_reads = max(1, min(int(G * cov / L), 40_000))
        starts = rng.integers(0, G - L, n_reads)
        reads = []
        for a in starts:
            r = genome[a:a + L].copy()
            if rng.random() < 0.5: r = (3 - r)[::-1]
            reads.append(r)
        if rng.random() < 0.3:
            reads += [reads[int(rng.integers(0, len(reads)))]] * int(rng.choice([100, 3000, 70_000]))
        sets.append(fastq_text(reads))
    try:
        for M in (1, 2, 3):
            b = K.Batch()
            for fq in sets: b.add_fastq(fq, Q=0)
            off, ids = ctx.sketch_batch(b, K.SKETCH_KEEP_ZERO | K.SKETCH_NO_CAPACITY, min_occ=M)
            for g, fq in enumerate(sets):
                n_cases += 1
                want = np.sort(sk.fastq(fq, Q=0, M=M))
                if not np.array_equal(ids[int(off[g]):int(off[g + 1])], want):
                    bad += 1; print("seed", seed, "params", (k, s, l), "-n", M, "set", g, "ids differ", len(want), int(off[g + 1] - off[g]), flush=True)
            b.close()
        b = K.Batch()
        for fq in sets: b.add_reads(fq)
        off, ids, cnt = ctx.sketch_batch_pos(b, K.SKETCH_KEEP_ZERO | K.SKETCH_NO_CAPACITY | K.SKETCH_COUNTS)
        for g, fq in enumerate(sets):
            n_cases += 1
            wi, wc = sk.fastq_koc(fq)
            o = np.argsort(wi)
            lo, hi = int(off[g]), int(off[g + 1])
            if not (np.array_equal(ids[lo:hi], wi[o]) and np.array_equal(cnt[lo:hi], wc[o].astype(np.uint32))):
                bad += 1; print("seed", seed, "params", (k, s, l), "abundances of set", g, "differ", flush=True)
        b.close()
    except K.KssdError as e:
        bad += 1; print("seed", seed, "params", (k, s, l), "ERROR", e, flush=True)
    ctx.close()
print("cases", n_cases, "bad", bad)
