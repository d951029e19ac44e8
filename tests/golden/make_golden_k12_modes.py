#!/usr/bin/env python3
"""Goldens for k - drlevel = 9 (-k 12 -s 6 -l 3: 36-bit tuples, 256 component files) in the modes beyond the plain one:
`-u` (uniq_fasta2co, iseq2comem.c:616-703), fastq `-n 2` (fastq2co, :277-356) and `-A` (mt_shortreads2koc + write_fqkoc2files,
:552-615,435-469) -- what the REAL reference binary (oracle/_ref/kssd, a 4 GiB table per run) writes for
tests/synth.py k12_mode_inputs.  Run in the dev container only:

    python tests/golden/make_golden_k12_modes.py

Writes k12_modes.npz: per mode the 256 component files (ids; for -A the abundances beside them) and their index files.
"""
import os
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import kssd_oracle as ko  # noqa: E402
import public_kssd_amd as K  # noqa: E402
from synth import k12_mode_inputs  # noqa: E402

K12 = (12, 6, 3)
SEED = 20260312
MODES = {"u": (["-u"], "dup.fa"), "n2": (["-n", "2"], "reads.fq"), "A": (["-A"], "reads.fq")}


def main():
    assert ko.have_ref(), "oracle/_ref/kssd missing: run `make -C oracle` in the dev container"
    tmp = tempfile.mkdtemp(prefix="kssd_golden_k12m_")
    res = {}
    try:
        K.Shuf.generate(*K12, seed=SEED).write(os.path.join(tmp, "k12.shuf"))
        for name, t in k12_mode_inputs().items():
            open(os.path.join(tmp, name), "wb").write(t)
        for tag, (extra, inp) in MODES.items():
            ko.run_ref(["dist", "-p", "1", "-L", "k12.shuf"] + extra + ["-o", "db_" + tag, inp], cwd=tmp, timeout=3600)
            d = os.path.join(tmp, "db_" + tag)
            stat = open(os.path.join(d, "cofiles.stat"), "rb").read()
            n = int(np.frombuffer(stat[20:24], np.int32)[0])
            assert n == 1 and int(np.frombuffer(stat[16:20], np.int32)[0]) == 256
            res[tag + ".stat"] = np.frombuffer(stat[:32 + 4 * n], np.uint8)
            tot = 0
            for c in range(256):
                co = np.fromfile(os.path.join(d, "combco.%d" % c), np.uint32)
                res["%s.co.%d" % (tag, c)] = co
                res["%s.idx.%d" % (tag, c)] = np.fromfile(os.path.join(d, "combco.index.%d" % c), np.uint64)
                tot += len(co)
                if tag == "A":
                    res["A.a.%d" % c] = np.fromfile(os.path.join(d, "combco.%d.a" % c), np.uint16)
                    assert len(res["A.a.%d" % c]) == len(co)
            print(tag, "tuples", tot, "sketch size in stat", np.frombuffer(stat[32:36], np.uint32)[0])
        np.savez_compressed(os.path.join(HERE, "k12_modes.npz"), **res)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
