"""The bit tricks of the device tokeniser (csrc/kssd_tok.inc, round 6) restated in Python against plain loops: `tok_hdr_masks` and
`tok_after_break` (markers resolved by the carries of one addition instead of doubling scans), `tok_emit_piece` (a thread's bases leave
as runs: the 2-bit codes of four bytes by one multiplication, a run to its place by one shift) and `tok_masks`' transposed gather.
The kernels themselves are held against the host tokeniser bit for bit in tests/test_gpu_tokenise.py (-m gpu); this pins the
reasoning on the CPU.  What is restated: iseq2comem.c:213-242 -- a '>' opens a header that the next line end closes, a base that
follows a run-breaking byte (not a line end) gets one invalid position in front."""
import random

M32 = 0xFFFFFFFF


def last_marker(setb, clr):
    """bit i: the nearest marker at or below i is a `set` bit (the doubling scan of rounds 2-5, byte by byte)"""
    out, state = 0, None
    for i in range(16):
        if (setb >> i) & 1:
            state = 1
        elif (clr >> i) & 1:
            state = 0
        if state == 1:
            out |= 1 << i
    return out


def hdr_masks(gt, nl):
    a = ~nl & 0xFFFF
    hdr1 = ((((a + gt) & M32) ^ a) | gt) & a
    mk = gt | nl
    open_ = (((mk & (-mk & M32)) - 1) & M32) & 0xFFFF
    return hdr1, open_


def after_break_loop(be, br):
    out, last = 0, None
    for i in range(16):
        if (be >> i) & 1:
            if last == "r":
                out |= 1 << i
            last = "b"
        elif (br >> i) & 1:
            last = "r"
    return out


def after_break(be, br):
    return (((~be) & 0xFFFF) + br) & be


def emit_loop(codes, be, br, st):
    """the byte loop of rounds 2-5: position 0 of the thread in bits 63-62 of img, mask bit lp per position"""
    ab = after_break_loop(be, br)
    first = be & -be
    ex = ab & ~first
    if (st & 4) and ((ab & first) or (st & 2)):
        ex |= first
    img = msk = lp = 0
    for i in range(16):
        lp += (ex >> i) & 1
        if (be >> i) & 1:
            img |= (codes[i] & 3) << (62 - 2 * lp)
            msk |= 1 << lp
            lp += 1
    return img, msk


def emit_runs(codes, be, br, st):
    packed = 0
    for j in range(4):                                   # four bases per byte by ONE multiplication per word
        c = codes[4 * j] | codes[4 * j + 1] << 8 | codes[4 * j + 2] << 16 | codes[4 * j + 3] << 24
        packed |= (((c * 0x40100401) & M32) >> 24) << (24 - 8 * j)
    img = msk = lp = 0
    rem, bp, pp = be, (st >> 2) & 1, (st >> 1) & 1
    while rem:
        low = rem & -rem
        s = low.bit_length() - 1
        rest = rem & (rem + low)
        run = rem ^ rest
        n = bin(run).count("1")
        below = low - 1
        ev = (be | br) & below
        nb = ((br >> (ev.bit_length() - 1)) & 1) if ev else 0
        lp += nb if (be & below) else (bp & (nb | pp))
        chunk = ((packed << (2 * s)) & M32) & ((M32 << (32 - 2 * n)) & M32)
        img |= (chunk << 32) >> (2 * lp)
        msk |= ((1 << n) - 1) << lp
        lp += n
        rem = rest
    return img, msk


def test_markers_by_carries_equal_the_scans():
    rng = random.Random(1)
    for _ in range(100_000):
        gt = rng.getrandbits(16) & rng.getrandbits(16)
        nl = rng.getrandbits(16) & rng.getrandbits(16) & ~gt
        hdr1, open_ = hdr_masks(gt, nl)
        assert hdr1 == last_marker(gt, nl)
        mk, want = gt | nl, 0
        for i in range(16):
            if mk & ((1 << (i + 1)) - 1) == 0:
                want |= 1 << i
        assert open_ == want
        be = rng.getrandbits(16)
        br = rng.getrandbits(16) & ~be
        if rng.random() < 0.5:
            br &= rng.getrandbits(16)
        assert after_break(be, br) == after_break_loop(be, br)


def test_bases_leaving_as_runs_equal_the_byte_loop():
    rng = random.Random(2)
    for _ in range(100_000):
        be = rng.getrandbits(16)
        if rng.random() < 0.5:
            be |= rng.getrandbits(16)                    # (sequence text: long runs of bases)
        br = rng.getrandbits(16) & ~be
        if rng.random() < 0.6:
            br &= rng.getrandbits(16) & rng.getrandbits(16)
        st = rng.getrandbits(3)
        codes = [rng.getrandbits(2) for _ in range(16)]
        assert emit_loop(codes, be, br, st) == emit_runs(codes, be, br, st), (be, br, st)


def test_class_flags_gathered_from_transposed_words():
    rng = random.Random(3)

    def perm(hi, lo, sel):
        pool = [(lo >> (8 * i)) & 255 for i in range(4)] + [(hi >> (8 * i)) & 255 for i in range(4)]
        return sum(pool[(sel >> (8 * i)) & 255] << (8 * i) for i in range(4))
    for _ in range(20_000):
        by = [rng.getrandbits(8) for _ in range(16)]
        w = [by[4 * a] | by[4 * a + 1] << 8 | by[4 * a + 2] << 16 | by[4 * a + 3] << 24 for a in range(4)]
        a0, a1 = perm(w[1], w[0], 0x05010400), perm(w[1], w[0], 0x07030602)
        b0, b1 = perm(w[3], w[2], 0x05010400), perm(w[3], w[2], 0x07030602)
        t = [perm(b0, a0, 0x05040100), perm(b0, a0, 0x07060302), perm(b1, a1, 0x05040100), perm(b1, a1, 0x07060302)]
        for j in range(4):
            for k in range(4):
                assert (t[j] >> (8 * k)) & 255 == by[4 * k + j]
        f = 0
        for j in range(4):
            f |= (t[j] & 0x80808080) >> (7 - j)           # (the class flag: here the bytes' top bits)
        u = f | (f >> 4)
        got = (u & 0xFF) | ((u >> 8) & 0xFF00)
        assert got == sum(1 << i for i in range(16) if by[i] & 0x80)
