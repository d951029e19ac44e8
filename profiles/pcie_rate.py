#!/usr/bin/env python3
"""One-off measurement for DESIGN.md section 5: the PCIe-inclusive rate of the HOST-level entry point
kssd_gpu_sketch_batch (host packed buffers in, host CSR out: hipMalloc + H2D + kernels + D2H), next to the host
tokeniser's own rate.  Run on the GPU box:  python profiles/pcie_rate.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import public_kssd_amd as K  # noqa: E402
from synth import fasta_text  # noqa: E402

G, L = 128, 5_000_000
rng = np.random.default_rng(1)
texts = [fasta_text(rng.integers(0, 4, L, dtype=np.uint8)) for _ in range(G)]
shuf = K.Shuf.generate(10, 6, 3, seed=20260101)
ctx = K.GpuCtx(shuf, 0)
b = K.Batch()
t0 = time.perf_counter()
for t in texts:
    b.add_fasta(t)
t_tok = time.perf_counter() - t0
ctx.sketch_batch(b)  # warm-up: workspaces, first-touch
best = 1e9
for _ in range(3):
    t0 = time.perf_counter()
    off, ids = ctx.sketch_batch(b)
    best = min(best, time.perf_counter() - t0)
print("host tokeniser (1 thread): %d genomes in %.2f s = %.0f Mbase/s" % (G, t_tok, G * L / t_tok / 1e6))
print("kssd_gpu_sketch_batch, host buffers in / host CSR out: %d x %.0f Mb (%.0f MB packed+mask) in %.1f ms = %.0f genomes/s, "
      "%.1f GB/s of packed input" % (G, L / 1e6, 0.375 * G * L / 1e6, best * 1e3, G / best, 0.375 * G * L / best / 1e9))
ctx.close()
