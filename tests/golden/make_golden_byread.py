#!/usr/bin/env python3
"""Golden vectors of the by-read sketches (dist --byread, reads2mco): the REAL reference (oracle/_ref/kssd) run on
qry_fa/edge.fa and on byread.fa.gz, a multi-record FASTA this script also generates (seeded; records of 10..9000
bases, empty records, N runs, repeated records, bases in front of the first header).  Run in the dev container only:

    python tests/golden/make_golden_byread.py

Writes byread.fa.gz (input; the reference opens --byread inputs without zcat, so the tests unpack it first) and
byread.npz: for every (shuffle, input) the u32 stream of combco.<c> and the i64 cumulative index combco.index.<c> of
every component, and cofiles.stat's first 32 bytes.
"""
import gzip
import json
import os
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import kssd_oracle as ko  # noqa: E402
import public_kssd_amd as K  # noqa: E402

SHUFS = {"L3K10": (10, 6, 3), "L3K11": (11, 6, 3)}  # K11: 16 components


def make_input(seed):
    rng = np.random.default_rng(seed)

    def seq(n):
        return "".join("ACGT"[i] for i in rng.integers(0, 4, n))
    out = [seq(700) + "\n"]  # in front of the first header: read 0
    keep = []
    for i in range(90):
        s = seq(int(rng.integers(10, 9000)))
        if i % 11 == 5 and keep:
            s = keep[int(rng.integers(0, len(keep)))]  # a repeated record: its k-mers are written again
        keep.append(s)
        out.append(">read%d some text\n" % i)
        if i % 9 == 4:
            s = s[:len(s) // 2] + "NNNNNNNNNN" + s[len(s) // 2:]
        out += [s[j:j + 70] + "\n" for j in range(0, len(s), 70)]
        if i % 13 == 7:
            out.append(">empty%d\n" % i)
    return "".join(out).encode()


def main():
    assert ko.have_ref(), "oracle/_ref/kssd missing: run `make -C oracle` in the dev container"
    meta = json.load(open(os.path.join(HERE, "golden.json")))
    text = make_input(meta["seed"] + 17)
    with gzip.GzipFile(os.path.join(HERE, "byread.fa.gz"), "wb", mtime=0) as f:
        f.write(text)
    tmp = tempfile.mkdtemp(prefix="kssd_golden_byread_")
    res = {}
    try:
        open(os.path.join(tmp, "byread.fa"), "wb").write(text)
        shutil.copy(os.path.join(HERE, "qry_fa", "edge.fa"), os.path.join(tmp, "edge.fa"))
        for tag, (k, s, l) in SHUFS.items():
            K.Shuf.generate(k, s, l, seed=meta["seed"]).write(os.path.join(tmp, tag + ".shuf"))
            for inp in ("byread.fa", "edge.fa"):
                out = "%s_%s" % (tag, inp)
                ko.run_ref(["dist", "--byread", "-L", tag + ".shuf", "-o", out, inp], cwd=tmp)
                d = os.path.join(tmp, out)
                ncomp = len([f for f in os.listdir(d) if f.startswith("combco.index.")])
                tot = 0
                for c in range(ncomp):
                    res["%s/%s/co.%d" % (tag, inp, c)] = np.fromfile(os.path.join(d, "combco.%d" % c), np.uint32)
                    res["%s/%s/idx.%d" % (tag, inp, c)] = np.fromfile(os.path.join(d, "combco.index.%d" % c), np.int64)
                    tot += len(res["%s/%s/co.%d" % (tag, inp, c)])
                res["%s/%s/stat" % (tag, inp)] = np.frombuffer(open(os.path.join(d, "cofiles.stat"), "rb").read()[:32], np.uint8)
                print(tag, inp, "components", ncomp, "entries", tot, "reads", len(res["%s/%s/idx.0" % (tag, inp)]) - 1)
        np.savez_compressed(os.path.join(HERE, "byread.npz"), **res)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
