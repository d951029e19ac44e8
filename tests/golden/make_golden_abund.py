#!/usr/bin/env python3
"""Golden vectors of the abundance sketches (dist -A): the REAL reference (oracle/_ref/kssd) run with one thread on the
committed reads.fq.gz.  Run in the dev container only:

    python tests/golden/make_golden_abund.py

Writes abund.npz: the ids of combco.0 in file order, the u16 occurrences of combco.0.a, combco.index.0 and the koc byte
of cofiles.stat.  -p 1 because the reference's insertions race between threads (iseq2comem.c:570) and the file order
follows the insertion order.
"""
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


def main():
    assert ko.have_ref(), "oracle/_ref/kssd missing: run `make -C oracle` in the dev container"
    meta = json.load(open(os.path.join(HERE, "golden.json")))
    sh = meta["shuf"]
    tmp = tempfile.mkdtemp(prefix="kssd_golden_abund_")
    try:
        sp = os.path.join(tmp, "L3K10.shuf")
        K.Shuf.generate(sh["k"], sh["subk"], sh["drlevel"], seed=meta["seed"]).write(sp)
        ko.run_ref(["dist", "-p", 1, "-A", "-L", sp, "-o", "koc", os.path.join(HERE, "reads.fq.gz")], cwd=tmp)
        d = os.path.join(tmp, "koc")
        stat = open(os.path.join(d, "cofiles.stat"), "rb").read()
        ids = np.fromfile(os.path.join(d, "combco.0"), dtype=np.uint32)
        counts = np.fromfile(os.path.join(d, "combco.0.a"), dtype=np.uint16)
        index = np.fromfile(os.path.join(d, "combco.index.0"), dtype=np.uint64)
        assert len(ids) == len(counts) == int(index[-1]) and len(ids) > 0
        np.savez_compressed(os.path.join(HERE, "abund.npz"), ids=ids, counts=counts, index=index,
                            koc=np.uint8(stat[4]), stat_head=np.frombuffer(stat[:32], np.uint8))
        print("abund.npz:", len(ids), "ids, occurrences max", counts.max(), "sum", counts.sum(), "koc", stat[4])
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
