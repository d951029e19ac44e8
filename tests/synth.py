"""Seeded synthetic inputs shared by the tests and bench.py (BASELINE.md section 3 generator, scaled)."""
import numpy as np

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def fasta_text(codes, name=b"seq", width=70, n_mask=None):
    """codes: uint8 array of 0..3; n_mask: optional bool array, True -> 'N'"""
    s = ACGT[codes]
    if n_mask is not None:
        s = s.copy()
        s[n_mask] = ord("N")
    n = len(s)
    rows = (n + width - 1) // width
    buf = np.full((rows, width + 1), ord("\n"), dtype=np.uint8)
    flat = np.zeros(rows * width, dtype=np.uint8)
    flat[:n] = s
    buf[:, :width] = flat.reshape(rows, width)
    body = buf.reshape(-1)
    if rows:
        tail = n - (rows - 1) * width  # bases on the last line
        body = np.concatenate([body[:(rows - 1) * (width + 1) + tail], np.array([ord("\n")], dtype=np.uint8)])
    return b">" + name + b"\n" + body.tobytes()


def clade_genomes(n_clades, per_clade, length, seed, sub_lo=0.005, sub_hi=0.05, p_n=1e-4):
    """list of (name, codes uint8, n_mask bool): clade ancestors i.i.d. uniform, members carry per-base
    substitutions at a rate drawn uniformly in [sub_lo, sub_hi], plus a sprinkle of N"""
    rng = np.random.default_rng(seed)
    out = []
    for c in range(n_clades):
        anc = rng.integers(0, 4, length, dtype=np.uint8)
        for m in range(per_clade):
            rate = rng.uniform(sub_lo, sub_hi)
            mut = rng.random(length) < rate
            codes = anc.copy()
            codes[mut] = (codes[mut] + rng.integers(1, 4, int(mut.sum()), dtype=np.uint8)) & 3
            nm = rng.random(length) < p_n
            out.append((b"c%d_m%d" % (c, m), codes, nm))
    return out


def fastq_text(codes_list, qual=b"I", names=None):
    out = []
    for i, codes in enumerate(codes_list):
        s = bytes(ACGT[codes])
        nm = names[i] if names else b"r%d" % i
        out.append(b"@" + nm + b"\n" + s + b"\n+\n" + qual * len(s) + b"\n")
    return b"".join(out)
