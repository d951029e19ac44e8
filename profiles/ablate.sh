#!/bin/bash
# development helper (GPU box): scan-kernel time with parts of the kernel compiled out
cd $GRAFT_REPO_ROOT
for a in 1 2 3 0; do
  KSSD_DEV_ABLATE=$a timeout 120 python - <<PY
import sys, os, time, torch, numpy as np
sys.path.insert(0, '.')
import bench, public_kssd_amd as K
dev = torch.device('cuda', 0)
shuf = K.Shuf.generate(10, 6, 3, seed=20260101); ctx = K.GpuCtx(shuf, 0)
G, L = 400, 5_000_000
packed, mask, chunk_off, _ = bench.make_batch(G, L, 20, 1, dev)
cap = int(G * L / 4096 * 1.25) + 4096
off = torch.zeros(G + 1, dtype=torch.int64, device=dev); ids = torch.zeros(cap, dtype=torch.int32, device=dev)
s = torch.cuda.current_stream().cuda_stream
for _ in range(3): ctx.sketch_device(packed, mask, chunk_off, off, ids, cap, 0, 1, s)
torch.cuda.synchronize(); ctx.kernel_time(0, True)
for _ in range(10): ctx.sketch_device(packed, mask, chunk_off, off, ids, cap, 0, 1, s)
torch.cuda.synchronize(); ms, n = ctx.kernel_time(0)
print("ablate=%s scan %.3f ms  -> %.1f GB/s algorithmic (%d genomes)" % (os.environ.get('KSSD_DEV_ABLATE'), ms, 0.375 * G * L / ms / 1e6, G))
ctx.close()
PY
done
