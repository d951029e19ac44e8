"""What bounds `kssd dist` once start-up is amortised: stage I on N names hard-linked onto 128 distinct 5 Mb FASTA files in tmpfs, under
different worker / buffer settings.  GPU box only:  python profiles/e2e_steady_probe.py [names=8192] [out file]"""
import json, os, shutil, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import public_kssd_amd as K
from synth import fasta_text, host_cores
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
BIN = os.path.join(ROOT, "public_kssd_amd", "kssd")
d = tempfile.mkdtemp(prefix="kssd_probe_", dir="/dev/shm")
try:
    rng = np.random.default_rng(1)
    os.mkdir(os.path.join(d, "src")); os.mkdir(os.path.join(d, "fa"))
    for i in range(128):
        open(os.path.join(d, "src", "g%03d.fasta" % i), "wb").write(fasta_text(rng.integers(0, 4, 5_000_000, dtype=np.uint8), b"g%d" % i))
    for r in range(N // 128):
        for i in range(128):
            os.link(os.path.join(d, "src", "g%03d.fasta" % i), os.path.join(d, "fa", "h%03d_g%03d.fasta" % (r, i)))
    K.Shuf.generate(10, 6, 3, seed=20260101).write(os.path.join(d, "L3K10.shuf"))
    cores = host_cores()
    settings = [{}, {"KSSD_WORKERS_PER_DEVICE": "3"}, {"KSSD_WORKERS_PER_DEVICE": "4"}, {"KSSD_WORKERS_PER_DEVICE": "3", "KSSD_TEXT_BUFFERS_EXTRA": "3"},
                {"KSSD_TEXT_BUFFERS_EXTRA": "3"}, {}] + [dict(e.split("=") for e in a.split(",")) for a in sys.argv[3:]]
    for env in settings:
        for rep in range(2):
            shutil.rmtree(os.path.join(d, "o"), ignore_errors=True)
            t0 = time.time()
            r = subprocess.run([BIN, "dist", "-p", str(cores), "-L", "L3K10.shuf", "-o", "o", "fa"], cwd=d, env=dict(os.environ, KSSD_TIMING="1", **env),
                               stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
            dt = time.time() - t0
            st = [json.loads(l) for l in r.stderr.decode().splitlines() if l.startswith('{"kssd_timing": "stage1"')]
            s = st[0] if st else {}
            steady = s.get("s_total", 0) - s.get("s_context_create_max", 0) - s.get("s_assemble_write", 0)
            print(json.dumps({"env": env, "rc": r.returncode, "seconds": round(dt, 3), "genomes_per_s": round(N / dt), "steady_genomes_per_s": round(N / steady) if steady > 0 else None,
                              "ms_per_job": round(steady / max(1, s.get("batches", 1)) * 1e3, 3),
                              **{k: s.get(k) for k in ("batches", "s_total", "s_context_create_max", "s_copy_threads_summed", "s_wait_text_buffer", "s_workers_summed", "s_device_calls_summed", "s_assemble_write")}}), file=out, flush=True)
finally:
    shutil.rmtree(d, ignore_errors=True)
