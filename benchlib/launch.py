"""Launcher plumbing of the bench: host thread count under a cgroup quota, stderr logging, the self-launch of --gpus N under
torch.distributed.run."""
import os
import subprocess
import sys

import torch


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def host_cores():
    """host threads worth starting: the machine's processors, but not more than the CPU time the container may use
    (cgroup cpu.max): on the measurement box 256 processors are visible under a quota of 16"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, -(-int(q) // int(p))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, -(-q // p)))
        except Exception:
            pass
    return n


def self_launch(a, script):
    """`python bench.py --gpus N` without a launcher (WORLD_SIZE unset): start the N ranks ourselves, as children, BEFORE
    anything in this process touches the GPU (never an exec from a process that has initialised HIP), relay rank 0's one
    JSON line and the children's exit code.  Under `python -m torch.distributed.run ... bench.py --gpus N` WORLD_SIZE is set
    and this function is not reached.  script: the bench's own file."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(script)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    log("[bench] --gpus %d without a launcher: starting %s" % (a.gpus, " ".join(cmd[1:10])))
    r = subprocess.run(cmd, stdout=subprocess.PIPE, env=env)
    line = None
    for ln in r.stdout.decode(errors="replace").splitlines():
        if ln.startswith('{"metric"'):
            line = ln
        elif ln.strip():
            log(ln)
    if line is not None:
        print(line, flush=True)
    if r.returncode != 0:
        return r.returncode
    return 0 if line is not None else 3
