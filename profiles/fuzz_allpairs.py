"""development tool (GPU box): `kssd dist --allpairs` (one command: sketch, keep on the device, exchange, search, report) on random
directories of FASTA files -- empty, tiny and multi-record files among them -- as one device, as n = 2..9 ranks on one device
(KSSD_EXCHANGE_FAKE_RANKS) in the own-index partition and with the full index on every rank (KSSD_ALLPAIRS_FULL_INDEX=1), against the
two-command flow (`dist -o sk in`; `dist -r sk -o d sk`): sharedk_ct.dat and distance.out byte for byte.
python3 profiles/fuzz_allpairs.py [cases] [first seed]"""
import os, shutil, subprocess, sys, tempfile
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in ("", "tests", "oracle"):
    sys.path.insert(0, os.path.join(R, d))
import numpy as np
BIN = os.path.join(R, "public_kssd_amd", "kssd")
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = 0
def run(args, cwd, env=None):
    return subprocess.run([BIN] + [str(a) for a in args], cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=dict(os.environ, **(env or {})))
for case in range(n_cases):
    rng = np.random.default_rng(424_000 + seed0 + case)
    d = tempfile.mkdtemp(prefix="kssd_ap_", dir="/dev/shm")
    try:
        run(["shuffle", "-k", 10, "-s", 6, "-l", 3, "-o", "p", "--seed", 500 + case], d)
        os.mkdir(os.path.join(d, "in"))
        nf = int(rng.integers(1, 24))
        base = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 200_000)]
        for f in range(nf):
            n = int(rng.choice([0, 30, 5000, 60_000, int(rng.integers(1000, 200_000))]))
            s = base[:n].copy()
            if n: s[rng.integers(0, n, max(1, n // 50))] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, max(1, n // 50))]   # relatives of one another
            recs = int(rng.choice([1, 1, 3]))
            parts = np.array_split(s, recs)
            t = b"".join(b">f%02d_r%d\n" % (f, r) + bytes(p) + b"\n" for r, p in enumerate(parts))
            open(os.path.join(d, "in", "g%02d.fasta" % f), "wb").write(t)
        opts = []
        if rng.random() < 0.3: opts += ["-M", int(rng.integers(0, 2))]
        if rng.random() < 0.3: opts += ["-N", int(rng.integers(1, nf + 1))]
        tag = "case %d seed %d files %d %s" % (case, 424_000 + seed0 + case, nf, " ".join(str(x) for x in opts))
        r1 = run(["dist", "-p", 4, "-L", "p.shuf", "-o", "sk", "in"], d)
        r2 = run(["dist", "-p", 4, "-r", "sk", "--keepskf", "-o", "two"] + opts + ["sk"], d)
        if r1.returncode or r2.returncode:
            ra = run(["dist", "-p", 4, "-L", "p.shuf", "--allpairs", "--keepskf", "-o", "one"] + opts + ["in"], d)
            if (ra.returncode == 0):
                bad += 1; print(tag, "the two-command flow failed (%d, %d), the one-command flow did not" % (r1.returncode, r2.returncode), r2.stdout.decode(errors="replace")[-200:], flush=True)
            continue
        want = {f: open(os.path.join(d, "two", f), "rb").read() for f in ("sharedk_ct.dat", "distance.out")}
        n_ranks = int(rng.integers(2, 10))
        for name, env in (("one", {}), ("ranks%d" % n_ranks, {"KSSD_EXCHANGE_FAKE_RANKS": str(n_ranks)}),
                          ("ranks%d_full" % n_ranks, {"KSSD_EXCHANGE_FAKE_RANKS": str(n_ranks), "KSSD_ALLPAIRS_FULL_INDEX": "1"}), ("one_full", {"KSSD_ALLPAIRS_FULL_INDEX": "1"})):
            ra = run(["dist", "-p", 4, "-L", "p.shuf", "--allpairs", "--keepskf", "-o", name] + opts + ["in"], d, env)
            if ra.returncode:
                bad += 1; print(tag, name, "FAILED", ra.returncode, ra.stdout.decode(errors="replace")[-300:].replace("\n", " "), flush=True); continue
            for f in want:
                if open(os.path.join(d, name, f), "rb").read() != want[f]:
                    bad += 1; print(tag, name, f, "DIFFERS from the two-command flow", flush=True); break
    finally:
        shutil.rmtree(d, ignore_errors=True)
print("cases", n_cases, "bad", bad)
