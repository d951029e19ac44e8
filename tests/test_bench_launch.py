"""bench.py --gpus N without a launcher starts its own ranks (python -m torch.distributed.run, 127.0.0.1) before anything in
the parent process touches the GPU, and relays their exit code.  Here (no GPU) every rank stops with the bench's own
"needs a GPU" message: the launch path itself is what is checked; the N > 1 measurement runs on the GPU box."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_starts_its_own_ranks():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the N > 1 line itself is produced by profiles/run_r03a.sh")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    err = r.stderr.decode(errors="replace")
    assert r.returncode != 0                                   # the children's failure is the parent's
    assert "starting -m torch.distributed.run" in err and "--nproc-per-node 2" in err
    assert err.count("bench.py needs a GPU") >= 2, err[-2000:]  # both ranks got as far as the bench's own check
    assert r.stdout.decode().strip() == ""                      # no line without a measurement
