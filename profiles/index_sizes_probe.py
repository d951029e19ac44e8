"""The index build at the bench's size (1.2 M ids) and at ten times it (12.2 M ids: 10 000 sketches -- configs[2], or the gathered
sketches of eight ranks), between events on the stream: partition + build per call.  KSSD_INDEX_ONE_LEVEL=1: the one-level partition."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import public_kssd_amd as K
dev = torch.device("cuda", 0)
rng = np.random.default_rng(2)
ctx = K.GpuCtx(kmerlen=20)
for G in (1000, 10000):
    S = 1222
    ids = torch.from_numpy(np.sort(rng.integers(0, 1 << 28, size=(G, S), dtype=np.int64).astype(np.uint32), axis=1).reshape(-1).view(np.int32)).to(dev)
    off = torch.arange(G + 1, dtype=torch.int64, device=dev) * S
    for env in ({}, {"KSSD_INDEX_ONE_LEVEL": "1"}):
        os.environ.pop("KSSD_INDEX_ONE_LEVEL", None)
        os.environ.update(env)
        for _ in range(5):
            ctx.index_build_device(off, ids, G, G * S, check=False)
        torch.cuda.synchronize()
        assert ctx.index_status() == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            ctx.index_build_device(off, ids, G, G * S, check=False)
        e1.record()
        torch.cuda.synchronize()
        print("%6d sketches, %9d ids %-28s index build %8.1f us" % (G, G * S, str(env), e0.elapsed_time(e1) / 30 * 1e3), flush=True)
