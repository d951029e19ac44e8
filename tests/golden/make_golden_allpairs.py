#!/usr/bin/env python3
"""All-pairs among the golden reference genomes (tests/golden/ref_fa), as the REAL reference binary (oracle/_ref/kssd)
computes it in its own three-command flow -- stage I + II of the directory as a database, stage I of the same directory
as the queries, the search with --keepskf:

    kssd dist -L L3K10.shuf -r ref_fa -o db ;  kssd dist -L L3K10.shuf -o co ref_fa ;  kssd dist -r db -o out --keepskf co

What it leaves (sharedk_ct.dat, by file name; distance.out) is what `kssd dist --allpairs` of this repository -- one command,
sketches resident on the devices -- must reproduce.  Run in the dev container only:

    python tests/golden/make_golden_allpairs.py
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
    ref_dir = os.path.join(HERE, "ref_fa")
    tmp = tempfile.mkdtemp(prefix="kssd_golden_ap_")
    try:
        sp = os.path.join(tmp, "L3K10.shuf")
        K.Shuf.generate(10, 6, 3, seed=meta["seed"]).write(sp)
        ko.run_ref(["dist", "-p", 2, "-L", sp, "-r", ref_dir, "-o", "db"], cwd=tmp)
        ko.run_ref(["dist", "-p", 2, "-L", sp, "-o", "co", ref_dir], cwd=tmp)
        texts = {}
        for tag, extra in (("M0_O2", []), ("M1_N3", ["-M", 1, "-N", 3])):
            ko.run_ref(["dist", "-p", 2, "-r", "db", "--keepskf"] + extra + ["-o", "out_" + tag, "co"], cwd=tmp)
            texts[tag] = open(os.path.join(tmp, "out_" + tag, "distance.out"), "rb").read().decode().replace(ref_dir, "FA")
        _, rsz, rnames = ko.read_stat(os.path.join(tmp, "db", "mcofiles.stat"), mco=True)
        _, qsz, qnames = ko.read_stat(os.path.join(tmp, "co", "cofiles.stat"))
        sh = np.fromfile(os.path.join(tmp, "out_M0_O2", "sharedk_ct.dat"), dtype=np.uint32).reshape(len(qnames), len(rnames))
        np.savez_compressed(os.path.join(HERE, "allpairs.npz"), shared=sh,
                            ref_names=np.array([os.path.basename(n) for n in rnames]),
                            qry_names=np.array([os.path.basename(n) for n in qnames]), ref_sz=rsz, qry_sz=qsz,
                            **{"distance_" + t: np.frombuffer(x.encode(), dtype=np.uint8) for t, x in texts.items()})
        print("all-pairs goldens written:", sh.shape, "\n" + texts["M0_O2"][:600])
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
