#!/usr/bin/env python3
"""development tool (GPU box): `kssd dist -L ... -o out <1 024 FASTA files in tmpfs>` under the pipeline's knobs -- sketch workers per
device (KSSD_WORKERS_PER_DEVICE) and text buffers beyond one per worker (KSSD_TEXT_BUFFERS_EXTRA) -- wall time and stage times,
several runs each.  The inputs are bench.py's own (128 distinct genomes of 5 Mb under 8 names each)."""
import json
import os
import shutil
import sys
import tempfile

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import public_kssd_amd as K

dev = torch.device("cuda", 0)
shuf = K.Shuf.generate(10, 6, 3, seed=20260101)
_, _, _, kept = bench.make_batch(128, 5_000_000, 50, 20260101, dev, keep_codes=128)
torch.cuda.synchronize()
cores = str(bench.host_cores())
base = "/dev/shm" if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > 12e9 else None
d = tempfile.mkdtemp(prefix="kssd_knobs_", dir=base)
try:
    shuf.write(os.path.join(d, "L3K10.shuf"))
    os.mkdir(os.path.join(d, "fa"))
    lut = bytes(b"ACGT")
    import numpy as np
    LUT = np.frombuffer(lut, dtype=np.uint8)
    paths = []
    for gi, c in enumerate(kept):
        seq = LUT[np.asarray(c)]
        p = os.path.join(d, "fa", "g%05d.fa" % gi)
        with open(p, "wb") as f:
            f.write((">g%d\n" % gi).encode())
            body = np.empty(len(seq) + (len(seq) + 69) // 70, dtype=np.uint8)
            # 70 bases per line
            nfull = len(seq) // 70
            lines = seq[:nfull * 70].reshape(nfull, 70)
            out = np.concatenate([lines, np.full((nfull, 1), 10, dtype=np.uint8)], axis=1).reshape(-1)
            f.write(out.tobytes())
            if len(seq) % 70:
                f.write(seq[nfull * 70:].tobytes() + b"\n")
        paths.append(p)
    for rep in range(1, 8):
        for gi, p in enumerate(paths):
            os.link(p, os.path.join(d, "fa", "g%05d_%d.fa" % (gi, rep)))
    n = len(os.listdir(os.path.join(d, "fa")))
    # one untimed run: the .core cache beside the .shuf, page cache
    bench._run_ours(["dist", "-p", cores, "-L", "L3K10.shuf", "-o", "warm", "fa"], d, {"KSSD_TIMING": "1"})
    shutil.rmtree(os.path.join(d, "warm"), ignore_errors=True)
    for wpd, extra in ((2, 1), (3, 1), (2, 3), (3, 3), (4, 2), (2, 1), (3, 1), (3, 3)):
        runs = []
        for r in range(3):
            dt, tm = bench._run_ours(["dist", "-p", cores, "-L", "L3K10.shuf", "-o", "o_%d_%d_%d" % (wpd, extra, r), "fa"], d,
                                     {"KSSD_TIMING": "1", "KSSD_WORKERS_PER_DEVICE": str(wpd), "KSSD_TEXT_BUFFERS_EXTRA": str(extra)})
            runs.append((dt, tm))
            shutil.rmtree(os.path.join(d, "o_%d_%d_%d" % (wpd, extra, r)), ignore_errors=True)
        runs.sort(key=lambda x: x[0])
        dt, tm = runs[0]
        print("workers/device %d  extra buffers %d: %d files  wall best %.3f s (runs %s)  -> %.0f genomes/s | s_total %.3f before_workers %.3f ctx %.3f read %.3f workers_summed %.3f device_calls %.3f write %.3f"
              % (wpd, extra, n, dt, " ".join("%.3f" % x[0] for x in runs), n / dt, tm["s_total"], tm["s_before_workers"], tm["s_context_create_max"], tm["s_read_gunzip"],
                 tm["s_workers_summed"], tm["s_device_calls_summed"], tm["s_assemble_write"]), flush=True)
finally:
    shutil.rmtree(d, ignore_errors=True)
