"""Does a plain fill of the distance planes (no LDS, few registers) hide under the scan?  Sketch steps of the bench batch alone,
then with N MB filled on a second stream during every step (torch fill_ kernels), then the fill alone.  GPU box."""
import sys, time, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import public_kssd_amd as K
from benchlib.workloads import make_batch

dev = torch.device("cuda", 0)
G, L = 1000, 5_000_000
shuf = K.Shuf.generate(10, 6, 3, seed=20260101)
packed, mask, chunk_off, _ = make_batch(G, L, 50, 20260101, dev)
ctx = K.GpuCtx(shuf, 0)
cap = int(G * L / 4096 * 1.25) + 4096
off = torch.zeros(G + 1, dtype=torch.int64, device=dev)
ids = torch.zeros(cap, dtype=torch.int32, device=dev)
main = torch.cuda.current_stream()
side = torch.cuda.Stream(device=dev)
for _ in range(3):
    ctx.sketch_device(packed, mask, chunk_off, off, ids, cap)
    print(ctx.sketch_status())


def run(n, mb, sketch=True, fill=True):
    bufs = [torch.empty(mb * 1024 * 1024 // 8 // 4, dtype=torch.float64, device=dev) for _ in range(4)] if mb else []
    ev0, ev1 = torch.cuda.Event(), torch.cuda.Event()
    def step():
        if mb and fill:
            ev0.record(main)
            side.wait_event(ev0)
            with torch.cuda.stream(side):
                for i, b in enumerate(bufs):
                    b.fill_(1.0 if i & 1 else 0.0)
                ev1.record(side)
        if sketch:
            ctx.sketch_device(packed, mask, chunk_off, off, ids, cap, stream=main.cuda_stream)
        if mb and fill:
            main.wait_event(ev1)
    for _ in range(40):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for rep in range(2):
    print("sketch alone            %.4f ms" % run(50, 0))
    for mb in (36, 288):
        print("fill of %3d MB alone    %.4f ms" % (mb, run(50, mb, sketch=False)))
        print("sketch + fill of %3d MB %.4f ms" % (mb, run(50, mb)))
