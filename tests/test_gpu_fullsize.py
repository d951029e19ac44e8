"""-m gpu: BASELINE configs[1] at FULL size (1 000 x 5 Mb, L3K10, all-pairs 1e6) through the device-level C ABI, checked
by properties that do not need the oracle on 5 Gbase -- sortedness, idempotence, a checksum of checksums of the
all-pairs matrix, symmetry, metric identities -- plus bit-exact oracle parity on a sample of the genomes."""
import os
import sys

import numpy as np
import pytest

import kssd_oracle as ko
import public_kssd_amd as K
from synth import fasta_text

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_config2_full_size_properties(shuf_l3k10):
    import torch
    sys.path.insert(0, ROOT)
    import bench
    dev = torch.device("cuda", 0)
    G, L, SAMPLE = 1000, 5_000_000, 12
    packed, mask, chunk_off, kept = bench.make_batch(G, L, 50, 20260101, dev, keep_codes=SAMPLE)
    ctx = K.GpuCtx(shuf_l3k10, 0)
    try:
        cap = int(G * L / 4096 * 1.25) + 4096
        outs = []
        for rep in range(2):
            off = torch.zeros(G + 1, dtype=torch.int64, device=dev)
            ids = torch.zeros(cap, dtype=torch.int32, device=dev)
            for attempt in range(6):
                ctx.sketch_device(packed, mask, chunk_off, off, ids, cap)
                rc, total, bad = ctx.sketch_status()
                if rc == 0:
                    break
                assert rc == K.capi.ERR_OVERFLOW, rc
            assert rc == 0
            outs.append((off, ids, int(total)))
        (off, ids, total), (off2, ids2, total2) = outs
        # idempotence: the same batch gives the same CSR, bit for bit
        assert total == total2 and torch.equal(off, off2) and torch.equal(ids[:total], ids2[:total])
        # sizes: ~ L / 4096 ids per genome (sampling rate 2^-12), no empty sketch
        sz = (off[1:] - off[:-1])
        assert int(off[0]) == 0 and int(off[-1]) == total
        assert 1000 < int(sz.min()) and int(sz.max()) < 1500 and abs(float(sz.double().mean()) - L / 4096) < 15
        # sortedness: ids strictly ascending inside every genome (=> distinct), all below 2^28
        v = ids[:total].to(torch.int64)
        inc = v[1:] > v[:-1]
        inc[(off[1:-1] - 1).clamp(min=0)] = True            # genome boundaries may go down
        assert bool(inc.all()) and int(v.max()) < (1 << 28) and int(v.min()) >= 1
        # oracle parity on a sample (the first genomes of the batch, as FASTA text)
        sk = ko.Sketcher(shuf_l3k10.table, 10, 6, 3)
        oh, ih = off.cpu().numpy(), ids[:total].cpu().numpy().view(np.uint32)
        for g, (codes, nmask) in enumerate(kept):
            want = np.sort(sk.fasta(fasta_text(codes, b"g%d" % g, n_mask=nmask)))
            assert np.array_equal(ih[int(oh[g]):int(oh[g + 1])], want), g
        # all-pairs: index + rows with the four metric planes
        shared = torch.zeros(G * G, dtype=torch.int32, device=dev)
        planes = [torch.zeros(G * G, dtype=torch.float64, device=dev) for _ in range(4)]
        ctx.index_build_device(off, ids, G, total)
        ctx.dist_device(off, ids, G, 0, G, shared, *planes)
        torch.cuda.synchronize()
        S = shared.view(G, G).to(torch.int64)
        assert torch.equal(S, S.t()), "shared counts are symmetric"
        assert torch.equal(S.diagonal(), sz), "a sketch shares all of itself"
        # checksum of checksums: sum of the matrix = sum over distinct ids of (number of genomes holding it)^2,
        # and every row sums to the total posting length of its ids -- both computed without the rows kernel
        uniq, inv, cnt = torch.unique(v, return_inverse=True, return_counts=True)
        assert int(S.sum()) == int((cnt * cnt).sum())
        gid = torch.repeat_interleave(torch.arange(G, device=dev), sz)
        rows = torch.zeros(G, dtype=torch.int64, device=dev).index_add_(0, gid, cnt[inv])
        assert torch.equal(S.sum(1), rows)
        # clades: 50 x 20 members; within a clade genomes share a lot, across clades next to nothing
        blk = S.view(50, 20, 50, 20)
        within = torch.stack([blk[c, :, c, :] for c in range(50)])
        assert int(within.min()) > 60
        cross = S.clone()
        for c in range(50):
            cross[c * 20:(c + 1) * 20, c * 20:(c + 1) * 20] = 0
        assert int(cross.max()) <= 3
        # metric identities on the full planes: J = s / (X + Y - s), C = s / min(X, Y) as IEEE divisions; distances in [0, 1]
        J, MD, C, AD = [p.view(G, G) for p in planes]
        X = sz.view(1, G).double()
        Y = sz.view(G, 1).double()
        assert torch.equal(J, S.double() / (X + Y - S.double()))
        assert torch.equal(C, S.double() / torch.minimum(X, Y))
        assert bool(((MD >= 0) & (MD <= 1) & (AD >= 0) & (AD <= 1)).all())
        assert bool((MD.diagonal() == 0).all()) and bool((AD.diagonal() == 0).all())
        assert bool((MD[S == 0] == 1).all()) and bool((AD[S == 0] == 1).all())
        # Mash / Aaf against the host formula (oracle, host libm) on ALL 1e6 pairs: never more than 1 ulp apart
        # (north_star tolerance), J and C bit for bit
        szh = sz.cpu().numpy().astype(np.uint32)
        Sh = S.cpu().numpy().astype(np.uint32)
        oJ, oMD, oC, oAD = ko.metrics_batch(szh[None, :], szh[:, None], Sh, 20)

        def ulps(a, b):
            return np.abs(a.cpu().numpy().view(np.int64) - b.view(np.int64))
        assert ulps(J, oJ).max() == 0 and ulps(C, oC).max() == 0
        dm, da = ulps(MD, oMD), ulps(AD, oAD)
        assert dm.max() <= 1 and da.max() <= 1, (int(dm.max()), int(da.max()))
        nz = Sh > 0
        assert (dm[nz] == 0).mean() > 0.97 and (da[nz] == 0).mean() > 0.97
    finally:
        ctx.close()
