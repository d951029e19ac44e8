"""Synthetic inputs of the bench, generated and packed on the device (setup, untimed): genome batches (configs[1] / [2]), read sets
(configs[3]) and long records (configs[4]).  Nothing here is timed and nothing here touches the oracle."""
import numpy as np
import torch

import public_kssd_amd as K

READ_LEN = 150  # configs[3]: reads of 150 bp


def make_batch(n_genomes, length, n_clades, seed, dev, keep_codes=0, on_genome=None, keep_on_device=False):
    """returns packed int32[words+slack], mask int32[...], chunk_off uint64[n+1], kept [(codes u8, nmask bool)]
    on_genome(gi, codes u8 tensor, nmask bool tensor): called for every genome (device tensors, valid during the call);
    keep_on_device: `kept` holds device tensors instead of numpy arrays"""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    chunks = (length + K.CHUNK_BASES - 1) // K.CHUNK_BASES
    padded = chunks * K.CHUNK_BASES
    packed = torch.zeros(n_genomes * chunks * K.CHUNK_WORDS + 64, dtype=torch.int32, device=dev)
    mask = torch.zeros(n_genomes * chunks * K.CHUNK_MASKW + 64, dtype=torch.int32, device=dev)
    wsh = (30 - 2 * torch.arange(16, device=dev, dtype=torch.int64))
    msh = torch.arange(32, device=dev, dtype=torch.int64)
    per = (n_genomes + n_clades - 1) // n_clades
    kept = []
    gi = 0
    for c in range(n_clades):
        anc = torch.randint(0, 4, (length,), generator=g, device=dev, dtype=torch.uint8)
        for m in range(per):
            if gi >= n_genomes:
                break
            rate = 0.005 + 0.045 * float(torch.rand((), generator=g, device=dev))
            mut = torch.rand(length, generator=g, device=dev) < rate
            add = torch.randint(1, 4, (length,), generator=g, device=dev, dtype=torch.uint8)
            codes = torch.where(mut, (anc + add) & 3, anc)
            nmask = torch.rand(length, generator=g, device=dev) < 1e-4
            valid = ~nmask
            codes_v = torch.where(valid, codes, torch.zeros_like(codes))
            cp = torch.zeros(padded, dtype=torch.int64, device=dev)
            cp[:length] = codes_v
            vp = torch.zeros(padded, dtype=torch.int64, device=dev)
            vp[:length] = valid
            w = (cp.view(-1, 16) << wsh).sum(1)
            mw = (vp.view(-1, 32) << msh).sum(1)
            packed[gi * chunks * K.CHUNK_WORDS:(gi + 1) * chunks * K.CHUNK_WORDS] = w.to(torch.int32)
            mask[gi * chunks * K.CHUNK_MASKW:(gi + 1) * chunks * K.CHUNK_MASKW] = mw.to(torch.int32)
            if gi < keep_codes:
                kept.append((codes, nmask) if keep_on_device else (codes.cpu().numpy(), nmask.cpu().numpy()))
            if on_genome is not None:
                on_genome(gi, codes, nmask)
            gi += 1
    chunk_off = np.arange(n_genomes + 1, dtype=np.uint64) * np.uint64(chunks)
    return packed, mask, chunk_off, kept


class FastaTextSink:
    """make_batch's genomes once more as FASTA TEXT on the device (a header line, 70 columns, 'N' where the generator put one): what
    `kssd dist` uploads for the same genomes -- the input of the device tokeniser's leg.  File g sits at a multiple of 16 bytes."""

    def __init__(self, n_genomes, length, dev, width=70):
        self.width, self.dev = width, dev
        rows = (length + width - 1) // width
        self.per = ((32 + length + rows) + 15) // 16 * 16
        self.text = torch.zeros(n_genomes * self.per + 64, dtype=torch.uint8, device=dev)
        self.off = np.zeros(n_genomes, dtype=np.uint64)
        self.len = np.zeros(n_genomes, dtype=np.uint64)
        self.lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)

    def __call__(self, gi, codes, nmask):
        L, w = int(codes.numel()), self.width
        ch = self.lut[codes.long()]
        ch = torch.where(nmask, torch.full_like(ch, ord("N")), ch)
        hdr = torch.tensor(list(b">g%d synthetic\n" % gi), dtype=torch.uint8, device=self.dev)
        full = L // w
        body = torch.cat([ch[:full * w].view(full, w), torch.full((full, 1), 10, dtype=torch.uint8, device=self.dev)], dim=1).reshape(-1)
        parts = [hdr, body]
        if L % w:
            parts += [ch[full * w:], torch.tensor([10], dtype=torch.uint8, device=self.dev)]
        t = torch.cat(parts)
        at = gi * self.per
        assert t.numel() <= self.per
        self.text[at:at + t.numel()] = t
        self.off[gi], self.len[gi] = at, t.numel()


def mask_summary(ctx, mask, n_chunks, dev):
    """the batch's summary words (include/kssd_gpu.h: one 64-bit word per chunk, bit l = lane l's 64 positions are all bases), written
    by the library's own kernel when the batch is made resident -- untimed setup like the packing itself; returns (tensor, lanes
    whose bit is clear = whose 8 bytes of mask the scan still fetches)"""
    summ = torch.zeros(max(n_chunks, 1), dtype=torch.int64, device=dev)
    ctx.mask_summarise_device(mask, n_chunks, summ)
    torch.cuda.synchronize()
    return summ


def summary_clear_lanes(summ, n_chunks):
    h = summ[:n_chunks].cpu().numpy().view(np.uint8)
    return int(n_chunks) * 64 - int(np.unpackbits(h).sum())


def make_reads_batch(src_codes, n_reads, seed, dev, err=0.005, keep_reads=0, slice_reads=1 << 21):
    """packed / mask / chunk_off of n_reads x 150 bp drawn from the device code tensors `src_codes` (equally long),
    either strand, substitution errors at rate err; the codes of the first keep_reads reads come back as a host array"""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    L = int(src_codes[0].numel())
    flat = torch.cat(src_codes)
    stride = READ_LEN + 1
    n_pos = n_reads * stride
    chunks = (n_pos + K.CHUNK_BASES - 1) // K.CHUNK_BASES
    packed = torch.zeros(chunks * K.CHUNK_WORDS + 64, dtype=torch.int32, device=dev)
    mask = torch.zeros(chunks * K.CHUNK_MASKW + 64, dtype=torch.int32, device=dev)
    wsh = (30 - 2 * torch.arange(16, device=dev, dtype=torch.int64))
    msh = torch.arange(32, device=dev, dtype=torch.int64)
    j = torch.arange(stride, device=dev, dtype=torch.int64)
    validj = (j < READ_LEN)
    kept = np.zeros((keep_reads, READ_LEN), dtype=np.uint8)
    for r0 in range(0, n_reads, slice_reads):
        S = min(slice_reads, n_reads - r0)
        S32 = (S + 31) // 32 * 32                       # whole mask words per slice (the surplus reads are cut off below)
        gi = torch.randint(0, len(src_codes), (S32,), generator=g, device=dev)
        st = torch.randint(0, L - READ_LEN, (S32,), generator=g, device=dev)
        rev = torch.rand(S32, generator=g, device=dev) < 0.5
        jj = torch.where(rev[:, None], (READ_LEN - 1 - j).clamp(min=0)[None, :], j.clamp(max=READ_LEN - 1)[None, :])
        v = flat[(gi * L + st)[:, None] + jj]
        v = torch.where(rev[:, None], 3 - v, v)
        e = torch.rand(v.shape, generator=g, device=dev) < err
        v = torch.where(e, (v + torch.randint(1, 4, v.shape, generator=g, device=dev, dtype=torch.uint8)) & 3, v)
        ok = validj[None, :] & (torch.arange(S32, device=dev) < S)[:, None]
        v = torch.where(ok, v, torch.zeros_like(v))
        if r0 < keep_reads:
            m = min(keep_reads - r0, S)
            kept[r0:r0 + m] = v[:m, :READ_LEN].cpu().numpy()
        w = (v.reshape(-1, 16).to(torch.int64) << wsh).sum(1).to(torch.int32)
        mw = (ok.reshape(-1, 32).to(torch.int64) << msh).sum(1).to(torch.int32)
        p0 = r0 * stride
        assert p0 % 32 == 0
        nw = min(len(w), (n_pos - p0 + 15) // 16)
        nm = min(len(mw), (n_pos - p0 + 31) // 32)
        packed[p0 // 16:p0 // 16 + nw] = w[:nw]
        mask[p0 // 32:p0 // 32 + nm] = mw[:nm]
        del v, e, ok, w, mw, jj
    return packed, mask, np.array([0, chunks], dtype=np.uint64), kept


def make_long_records(n, length, seed, dev, keep_first):
    """n records of `length` uniform random bases with 1e-5 isolated N, packed on the device slice by slice;
    returns packed, mask, chunk_off and (record 0's codes u8, N mask bool) on the host when keep_first"""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    chunks = (length + K.CHUNK_BASES - 1) // K.CHUNK_BASES
    packed = torch.zeros(n * chunks * K.CHUNK_WORDS + 64, dtype=torch.int32, device=dev)
    mask = torch.zeros(n * chunks * K.CHUNK_MASKW + 64, dtype=torch.int32, device=dev)
    wsh = (30 - 2 * torch.arange(16, device=dev, dtype=torch.int64))
    msh = torch.arange(32, device=dev, dtype=torch.int64)
    S = 1 << 26
    kept_c = np.empty(length, dtype=np.uint8) if keep_first else None
    kept_n = np.empty(length, dtype=bool) if keep_first else None
    for gi in range(n):
        for p0 in range(0, length, S):
            m = min(S, length - p0)
            mp = (m + 31) // 32 * 32
            codes = torch.randint(0, 4, (mp,), generator=g, device=dev, dtype=torch.uint8)
            isn = torch.rand(mp, generator=g, device=dev) < 1e-5
            isn[1:] &= ~isn[:-1]                                   # isolated: one N = one invalid position, as the tokeniser lays it out
            ok = ~isn
            ok[m:] = False
            codes = torch.where(ok, codes, torch.zeros_like(codes))
            if keep_first and gi == 0:
                kept_c[p0:p0 + m] = codes[:m].cpu().numpy()
                kept_n[p0:p0 + m] = isn[:m].cpu().numpy()
            w = (codes.view(-1, 16).to(torch.int64) << wsh).sum(1).to(torch.int32)
            mw = (ok.view(-1, 32).to(torch.int64) << msh).sum(1).to(torch.int32)
            b0 = gi * chunks * K.CHUNK_BASES + p0
            packed[b0 // 16:b0 // 16 + len(w)] = w
            mask[b0 // 32:b0 // 32 + len(mw)] = mw
            del codes, isn, ok, w, mw
    chunk_off = np.arange(n + 1, dtype=np.uint64) * np.uint64(chunks)
    return packed, mask, chunk_off, (kept_c, kept_n)
