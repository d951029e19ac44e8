"""`kssd dist -r <sketches> -o out <sketches>` at 4 096 x 4 096 on the GPU box: wall time, the command's stage times with the
report written through mappings by all threads (the tree) and by one thread's fwrite() (KSSD_REPORT_FWRITE=1).  python3 profiles/search4k_probe.py [runs]"""
import json, os, subprocess, sys, tempfile, time, shutil
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import public_kssd_amd as K
from benchlib.workloads import make_batch
BIN = os.path.join(ROOT, "public_kssd_amd", "kssd")
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
d = tempfile.mkdtemp(prefix="kssd_s4k_", dir="/dev/shm")
try:
    rng = np.random.default_rng(5)
    os.mkdir(os.path.join(d, "fa"))
    for i in range(128):
        a = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 1_000_000)]
        for r in range(32):
            b = a.copy(); m = rng.integers(0, len(b), 2000 * (r + 1)); b[m] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, len(m))]
            open(os.path.join(d, "fa", "g%03d_%02d.fasta" % (i, r)), "wb").write(b">g\n" + b.tobytes() + b"\n")
    K.Shuf.generate(10, 6, 3, seed=20260101).write(os.path.join(d, "L3K10.shuf"))
    subprocess.run([BIN, "dist", "-p", "16", "-L", "L3K10.shuf", "-o", "sk", "fa"], cwd=d, check=True, stdout=subprocess.DEVNULL)
    import hashlib
    sums = set()
    for env in ({}, {"KSSD_REPORT_FWRITE": "1"}, {}, {"KSSD_REPORT_FWRITE": "1"}):
        for _ in range(runs):
            shutil.rmtree(os.path.join(d, "out"), ignore_errors=True)
            t0 = time.time()
            r = subprocess.run([BIN, "dist", "-p", "16", "-r", "sk", "-o", "out", "sk"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, KSSD_TIMING="1", **env))
            dt = time.time() - t0
            assert r.returncode == 0, r.stderr.decode()[-1000:]
            tm = {}
            for l in r.stderr.decode().splitlines():
                if l.startswith("{"):
                    j = json.loads(l); tm.update({k: round(v, 3) for k, v in j.items() if k.startswith("s_")})
            h = hashlib.md5()
            with open(os.path.join(d, "out", "distance.out"), "rb") as fh:
                for blk in iter(lambda: fh.read(1 << 24), b""): h.update(blk)
            sums.add(h.hexdigest())
            print(env, "wall %.3f" % dt, tm, "distance.out %d MB" % (os.path.getsize(os.path.join(d, "out", "distance.out")) >> 20), flush=True)
    print("distinct distance.out files:", len(sums))
finally:
    shutil.rmtree(d, ignore_errors=True)
