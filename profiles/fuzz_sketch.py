"""development tool (GPU box): kssd_gpu_sketch_batch on random batches (parameter sets, lengths around chunk and block borders,
N runs, tiny and empty genomes, repeats) against the oracle"""
import os
import sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in ("", "tests", "oracle"):
    sys.path.insert(0, os.path.join(R, d))
import numpy as np
import kssd_oracle as ko
import public_kssd_amd as K
from synth import fasta_text

PARAMS = [(10, 6, 3), (8, 5, 2), (9, 6, 3), (11, 6, 3), (10, 7, 5), (8, 4, 1), (12, 7, 4), (9, 5, 2)]
bad = 0
n_cases = 0
for pi, (k, s, l) in enumerate(PARAMS):
    shuf = K.Shuf.generate(k, s, l, seed=100 + pi)
    sk = ko.Sketcher(shuf.table, k, s, l)
    ctx = K.GpuCtx(shuf, 0)
    for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
        rng = np.random.default_rng((int(sys.argv[2]) if len(sys.argv) > 2 else 0) + 1000 * pi + seed)
        texts = []
        for g in range(int(rng.integers(1, 12))):
            kind = int(rng.integers(0, 6))
            n = int(rng.choice([0, 1, 15, 16, 17, 2 * k - 1, 2 * k, 4095, 4096, 4097, 16383, 16384, 16385, 65536, int(rng.integers(100, 400_000))]))
            codes = rng.integers(0, 4, n, dtype=np.uint8)
            if kind == 1 and n > 100:                       # tandem repeat
                codes = np.tile(codes[: int(rng.integers(1, 50))], n)[:n]
            nm = np.zeros(n, dtype=bool)
            if kind == 2 and n:
                nm[rng.integers(0, n, max(1, n // 500))] = True
            if kind == 3 and n > 50:                        # a long N run in the middle
                a = int(rng.integers(0, n - 10))
                nm[a:a + int(rng.integers(1, 9000))] = True
            texts.append(fasta_text(codes, b"g%d" % g, n_mask=nm if nm.any() else None))
        b = K.Batch()
        for t in texts:
            b.add_fasta(t)
        # four seeds in ten: small limits on the LDS sort send ordinary genomes down the parts / ranges / global-memory paths
        ctx.set_lds_sort_limit(int(rng.choice([64, 256, 1024, 4096])) if rng.random() < 0.4 else 0)
        try:
            off, ids = ctx.sketch_batch(b, K.SKETCH_FASTA | K.SKETCH_NO_CAPACITY)
        except K.KssdError as e:
            print("params", (k, s, l), "seed", seed, "error", e)
            bad += 1
            continue
        try:  # the same batch with first positions (8-byte keys: what the command line asks for) must list the same ids
            off2, ids2, pos2 = ctx.sketch_batch_pos(b, K.SKETCH_FASTA | K.SKETCH_NO_CAPACITY)
            n_cases += 1
            if not (np.array_equal(off2, off) and np.array_equal(ids2, ids)):
                bad += 1
                print("params", (k, s, l), "seed", seed, "first-position call differs", flush=True)
        except K.KssdError as e:
            print("params", (k, s, l), "seed", seed, "first-position call: error", e)
            bad += 1
        for g, t in enumerate(texts):
            if len(t):
                wi, wc = sk.fasta(t, with_comps=True)          # stored ids and their components: the device lists whole tuples
                cb = 4 * max(k - l - 7, 0)
                want = np.sort((wi.astype(np.uint64) << np.uint64(cb) | wc.astype(np.uint64)).astype(np.uint32))
            else:
                want = np.zeros(0, np.uint32)
            got = ids[int(off[g]):int(off[g + 1])]
            n_cases += 1
            if not np.array_equal(got, want):
                bad += 1
                print("params", (k, s, l), "seed", seed, "genome", g, "len", len(t), "got", len(got), "want", len(want), flush=True)
        b.close()
    ctx.close()
print("cases", n_cases, "bad", bad)
