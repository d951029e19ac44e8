"""(The kernel this measured -- dist_flat_kernel -- is not in the product: measured, not kept; this script ran on the tree of the
commit before \"the other devices' rows: flat walk measured, not kept\".)
What the flat walk of the other devices' rows costs, by parts: the product kernel, the same with non-temporal id loads,
without the filter read (wrong results: the floor), and with other grids.  One library per run (KSSD_GPU_LIB)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import public_kssd_amd as K

dev = torch.device("cuda", 0)
rng = np.random.default_rng(1)
G, W, S = 1000, 8, 1220
def csr(n):
    ids = np.sort(rng.integers(0, 1 << 28, size=(n, S), dtype=np.int64).astype(np.uint32), axis=1)
    return np.arange(n + 1, dtype=np.int64) * S, ids.reshape(-1).view(np.int32)
ro, ri = csr(G)
qo, qi = csr(G * (W - 1))
d = [torch.from_numpy(x).to(dev) for x in (ro, ri, qo, qi)]
ctx = K.GpuCtx(kmerlen=20)
ctx.index_set_filter(True, 0, 0)
ctx.index_build_device(d[0], d[1], G, len(ri))
counts = torch.zeros(G * (W - 1) * G, dtype=torch.int32, device=dev)
def run(n=30):
    for _ in range(5):
        ctx.dist_counts_device(d[2], d[3], G * (W - 1), 0, G * (W - 1), counts)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        ctx.dist_counts_device(d[2], d[3], G * (W - 1), 0, G * (W - 1), counts)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
tag = os.environ.get("KSSD_GPU_LIB", "product").split("_")[-1]
for grid in ("", "512", "1024", "2048", "4096"):
    if grid:
        os.environ["KSSD_FLAT_GRID"] = grid
    print("%-12s grid %-6s flat (memset + kernel): %7.1f us   counts sum %d" % (tag, grid or "auto", run(), int(counts.sum().item())))
os.environ.pop("KSSD_FLAT_GRID", None)
os.environ["KSSD_DIST_NO_FLAT"] = "1"
print("%-12s one workgroup per row:          %7.1f us" % (tag, run()))
