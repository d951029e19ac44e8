"""End to end on the GPU box: `kssd dist` on 1 024 FASTA files (and the same gzip'ed) in tmpfs under tuning knobs of the host pipeline
with the cgroup's CPU accounting around it: is the command throttled by the box's 16-CPU quota (OpenMP threads that spin between parallel regions count as running).  Prints wall seconds and the command's own stage times."""
import json, os, subprocess, sys, tempfile, time, shutil, zlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import public_kssd_amd as K
BIN = os.path.join(ROOT, "public_kssd_amd", "kssd")
rng = np.random.default_rng(5)
d = tempfile.mkdtemp(prefix="kssd_e2e_", dir="/dev/shm")
try:
    os.mkdir(os.path.join(d, "fa")); os.mkdir(os.path.join(d, "gz"))
    for i in range(128):
        a = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 5_000_000)]
        buf = np.full((5_000_000 // 70 + 1, 71), 10, np.uint8)
        flat = np.full(buf.shape[0] * 70, ord("A"), np.uint8); flat[:5_000_000] = a
        buf[:, :70] = flat.reshape(-1, 70)
        t = b">g%d\n" % i + buf.tobytes()
        open(os.path.join(d, "fa", "r00_g%04d.fasta" % i), "wb").write(t)
        co = zlib.compressobj(1, zlib.DEFLATED, 31)
        open(os.path.join(d, "gz", "r00_g%04d.fasta.gz" % i), "wb").write(co.compress(t) + co.flush())
        for r in range(1, 8):
            os.symlink("r00_g%04d.fasta" % i, os.path.join(d, "fa", "r%02d_g%04d.fasta" % (r, i)))
            os.symlink("r00_g%04d.fasta.gz" % i, os.path.join(d, "gz", "r%02d_g%04d.fasta.gz" % (r, i)))
    K.Shuf.generate(10, 6, 3, seed=20260101).write(os.path.join(d, "L3K10.shuf"))
    def cpu_stat():
        try:
            return {l.split()[0]: int(l.split()[1]) for l in open("/sys/fs/cgroup/cpu.stat")}
        except OSError:
            return {}
    print("cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else None, "cpus", os.cpu_count(), flush=True)
    def run(src, env, p=16):
        out = os.path.join(d, "out"); shutil.rmtree(out, ignore_errors=True)
        c0 = cpu_stat(); t0 = time.time()
        r = subprocess.run([BIN, "dist", "-p", str(p), "-L", "L3K10.shuf", "-o", "out", src], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, KSSD_TIMING="1", **env))
        t1 = time.time(); c1 = cpu_stat()
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        tm = [json.loads(l) for l in r.stderr.decode().splitlines() if '"stage1"' in l][0]
        return round(t1 - t0, 3), {k: c1[k] - c0[k] for k in c1 if k in ("usage_usec", "nr_throttled", "throttled_usec")}, {k: round(v, 3) for k, v in tm.items() if k in ("s_total", "s_context_create_max", "s_read_gunzip", "s_device_calls_summed")}
    run("fa", {})
    for src in ("fa", "gz"):
        for env, p in (({}, 16), ({"OMP_WAIT_POLICY": "active"}, 16), ({"KSSD_DEV_SYNC_FLAGS": "4"}, 16), ({"KSSD_DEV_SYNC_FLAGS": "2"}, 16), ({}, 16), ({"OMP_WAIT_POLICY": "active"}, 16)):
            for _ in range(4):
                print(src, env, "-p", p, *run(src, env, p), flush=True)
finally:
    shutil.rmtree(d, ignore_errors=True)
