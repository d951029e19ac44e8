"""`kssd dist` on 1 024 FASTA files under rocprofv3 (kernel + memory-copy trace): what the device worker's 4 ms per job of 16 files are
made of.  Run on the GPU box: python3 profiles/e2e_trace.py <outdir>"""
import csv, glob, os, subprocess, sys, tempfile, shutil, collections
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import public_kssd_amd as K
BIN = os.path.join(ROOT, "public_kssd_amd", "kssd")
outdir = os.path.abspath(sys.argv[1])
rng = np.random.default_rng(5)
d = tempfile.mkdtemp(prefix="kssd_e2e_", dir="/dev/shm")
try:
    os.mkdir(os.path.join(d, "fa"))
    for i in range(128):
        a = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 5_000_000)]
        buf = np.full((5_000_000 // 70 + 1, 71), 10, np.uint8)
        flat = np.full(buf.shape[0] * 70, ord("A"), np.uint8); flat[:5_000_000] = a
        buf[:, :70] = flat.reshape(-1, 70)
        open(os.path.join(d, "fa", "r00_g%04d.fasta" % i), "wb").write(b">g%d\n" % i + buf.tobytes())
        for r in range(1, 8):
            os.symlink("r00_g%04d.fasta" % i, os.path.join(d, "fa", "r%02d_g%04d.fasta" % (r, i)))
    K.Shuf.generate(10, 6, 3, seed=20260101).write(os.path.join(d, "L3K10.shuf"))
    env = dict(os.environ, KSSD_SLOW_EXIT="1", KSSD_TIMING="1")
    subprocess.run([BIN, "dist", "-p", "16", "-L", "L3K10.shuf", "-o", "out0", "fa"], cwd=d, env=env, check=True, stdout=subprocess.DEVNULL)
    r = subprocess.run(["rocprofv3", "--kernel-trace", "--memory-copy-trace", "--output-format", "csv", "-d", os.path.join(outdir, "tr"), "--",
                        BIN, "dist", "-p", "16", "-L", "L3K10.shuf", "-o", "out1", "fa"], cwd=d, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    print(r.stderr.decode()[-1500:])
    kt = glob.glob(os.path.join(outdir, "tr", "**", "*kernel_trace.csv"), recursive=True)[0]
    mt = glob.glob(os.path.join(outdir, "tr", "**", "*memory_copy_trace.csv"), recursive=True)[0]
    ev = []
    agg = collections.defaultdict(lambda: [0, 0.0])
    for row in csv.DictReader(open(kt)):
        s, e = int(row["Start_Timestamp"]), int(row["End_Timestamp"])
        ev.append((s, e, "K " + row["Kernel_Name"][:50]))
        a = agg["K " + row["Kernel_Name"][:60]]; a[0] += 1; a[1] += (e - s) / 1e3
    for row in csv.DictReader(open(mt)):
        s, e = int(row["Start_Timestamp"]), int(row["End_Timestamp"])
        name = "M " + row.get("Direction", "?")
        ev.append((s, e, name + " %s B" % row.get("Size", "?")))
        a = agg[name]; a[0] += 1; a[1] += (e - s) / 1e3
    for k, (n, us) in sorted(agg.items(), key=lambda x: -x[1][1]):
        print("%-64s n %5d total %10.1f us avg %9.1f us" % (k, n, us, us / n))
    ev.sort()
    t0 = ev[0][0]
    print("first device event to last: %.1f ms; busy (union): " % ((max(e for _, e, _ in ev) - t0) / 1e6), end="")
    busy, cur_s, cur_e = 0, None, None
    for s, e, _ in ev:
        if cur_e is None or s > cur_e:
            if cur_e is not None: busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else: cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    print("%.1f ms" % (busy / 1e6))
    # one job in the middle: the events between two consecutive big H2D copies
    big = [i for i, (s, e, n) in enumerate(ev) if n.startswith("M") and "HOST_TO_DEVICE" in n and int(n.split()[-2]) > 10_000_000]
    if len(big) > 40:
        i0, i1 = big[30], big[32]
        for s, e, n in ev[i0:i1]:
            print("  +%9.1f us  %9.1f us  %s" % ((s - ev[i0][0]) / 1e3, (e - s) / 1e3, n))
    shutil.rmtree(os.path.join(outdir, "tr"), ignore_errors=True)
finally:
    shutil.rmtree(d, ignore_errors=True)
