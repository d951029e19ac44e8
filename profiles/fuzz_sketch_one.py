"""one case of profiles/fuzz_sketch.py again (parameter-set index, seed), with the status trace of the development library:
KSSD_GPU_LIB=public_kssd_amd/libkssd_gpu_dev.so KSSD_DEV_TRACE=1 python3 profiles/fuzz_sketch_one.py 5 48"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in ("", "tests", "oracle"):
    sys.path.insert(0, os.path.join(R, d))
import numpy as np
import kssd_oracle as ko
import public_kssd_amd as K
from synth import fasta_text
PARAMS = [(10, 6, 3), (8, 5, 2), (9, 6, 3), (11, 6, 3), (10, 7, 5), (8, 4, 1), (12, 7, 4), (9, 5, 2)]
pi, seed = int(sys.argv[1]), int(sys.argv[2])
k, s, l = PARAMS[pi]
shuf = K.Shuf.generate(k, s, l, seed=100 + pi)
ctx = K.GpuCtx(shuf, 0)
rng = np.random.default_rng(1000 * pi + seed)
texts = []
for g in range(int(rng.integers(1, 12))):
    kind = int(rng.integers(0, 6))
    n = int(rng.choice([0, 1, 15, 16, 17, 2 * k - 1, 2 * k, 4095, 4096, 4097, 16383, 16384, 16385, 65536, int(rng.integers(100, 400_000))]))
    codes = rng.integers(0, 4, n, dtype=np.uint8)
    if kind == 1 and n > 100:
        u = int(rng.integers(1, 50))
        print("   unit", u, "".join("ACGT"[c] for c in codes[:u]))
        codes = np.tile(codes[:u], n)[:n]
    nm = np.zeros(n, dtype=bool)
    if kind == 2 and n:
        nm[rng.integers(0, n, max(1, n // 500))] = True
    if kind == 3 and n > 50:
        a = int(rng.integers(0, n - 10))
        nm[a:a + int(rng.integers(1, 9000))] = True
    texts.append(fasta_text(codes, b"g%d" % g, n_mask=nm if nm.any() else None))
    print("genome", g, "kind", kind, "bases", n, flush=True)
keep = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else list(range(len(texts)))
b = K.Batch()
for i, t in enumerate(texts):
    if i in keep: b.add_fasta(t)
try:
    off, ids = ctx.sketch_batch(b, K.SKETCH_FASTA | K.SKETCH_NO_CAPACITY)
    print("ok", len(ids), np.diff(off))
except K.KssdError as e:
    print("error", e)
