"""Host report from candidate pairs (kssd_distance_print_pairs) against the dense report -- CPU only: a candidate list that
holds every pair with shared > 0 (what the device selection returns for -N) or every pair (the trivial superset) must
give the dense report's text byte for byte, for every option set."""
import numpy as np
import pytest

import kssd_oracle as ko
import public_kssd_amd as K


@pytest.mark.parametrize("opts", [dict(metric=0, n_max=2), dict(metric=1, n_max=4, pfield=1), dict(metric=0, dthreshold=0.08),
                                  dict(metric=1, correction=1), dict(metric=0)])
def test_pairs_report_equals_dense_report(tmp_path, opts):
    rng = np.random.default_rng(3)
    pool = rng.choice(1 << 28, 3000, replace=False).astype(np.uint32)

    def sk(n, lo, hi):
        off, ids = [0], []
        for g in range(n):
            own = np.sort(rng.choice(pool[:1500] if g % 2 else pool[1500:], int(rng.integers(lo, hi)), replace=False))
            ids.append(own)
            off.append(off[-1] + len(own))
        return np.array(off, np.uint64), np.concatenate(ids)
    roff, rids = sk(14, 200, 600)
    qoff, qids = sk(9, 100, 500)
    shared = ko.shared_counts(roff, rids, qoff, qids)
    ref = K.SketchSet(9, 20, 6, 1, ["r%d" % i for i in range(14)], roff, rids)
    qry = K.SketchSet(9, 20, 6, 1, ["q%d" % i for i in range(9)], qoff, qids)
    a = str(tmp_path / "dense.out")
    K.distance_print(a, shared, ref, qry, **opts)
    keep = shared > 0 if opts.get("n_max") else np.ones_like(shared, bool)
    poff = np.concatenate([[0], np.cumsum(keep.sum(1))]).astype(np.uint64)
    q_idx, r_idx = np.nonzero(keep)
    b = str(tmp_path / "pairs.out")
    K.distance_print_pairs(b, poff, r_idx.astype(np.uint32), shared[q_idx, r_idx], ref, qry, **opts)
    assert open(a, "rb").read() == open(b, "rb").read()
