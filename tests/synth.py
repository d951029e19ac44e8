"""Seeded synthetic inputs shared by the tests and bench.py (BASELINE.md section 3 generator, scaled)."""
import numpy as np

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def host_cores():
    """threads worth starting: the processors this process may run on, but not more than the CPU time its cgroup grants
    (the measurement box shows 256 processors under a quota of 16: a pool of 256 oracle threads there is slow, not wrong)"""
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, -(-int(q) // int(p))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, -(-q // p)))
        except Exception:
            pass
    return n


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


def fastq_records(reads2d, qual=ord("I")):
    """FASTQ text of n equally long reads (uint8 codes [n, len]) with fixed-width names, built without a Python loop:
    @r0000000\\n<bases>\\n+\\n<qualities>\\n"""
    n, ln = reads2d.shape
    nd = max(7, len(str(max(n - 1, 0))))
    rec = 2 + nd + 1 + ln + 1 + 2 + ln + 1
    out = np.empty((n, rec), dtype=np.uint8)
    out[:, 0] = ord("@")
    out[:, 1] = ord("r")
    idx = np.arange(n, dtype=np.int64)
    for d in range(nd):
        out[:, 2 + d] = ord("0") + (idx // 10 ** (nd - 1 - d)) % 10
    c = 2 + nd
    out[:, c] = ord("\n")
    out[:, c + 1:c + 1 + ln] = ACGT[reads2d]
    c += 1 + ln
    out[:, c] = ord("\n")
    out[:, c + 1] = ord("+")
    out[:, c + 2] = ord("\n")
    out[:, c + 3:c + 3 + ln] = qual
    out[:, c + 3 + ln] = ord("\n")
    return out.tobytes()


def sample_reads(genomes, n_reads, read_len, seed, err=0.005):
    """uint8 codes [n_reads, read_len]: uniform start positions on uniformly chosen genomes (equally long uint8 code
    arrays), either strand, per-base substitution errors at rate err"""
    rng = np.random.default_rng(seed)
    L = len(genomes[0])
    flat = np.concatenate(genomes)
    g = rng.integers(0, len(genomes), n_reads)
    s = rng.integers(0, L - read_len, n_reads)
    rev = rng.random(n_reads) < 0.5
    j = np.arange(read_len, dtype=np.int64)
    first = g * L + s + np.where(rev, read_len - 1, 0)
    step = np.where(rev, -1, 1)
    r = np.empty((n_reads, read_len), dtype=np.uint8)
    B = 65536  # block-wise: large temporaries are expensive to fault in
    idx = np.empty((B, read_len), dtype=np.int64)
    for b0 in range(0, n_reads, B):
        b1 = min(n_reads, b0 + B)
        ix = idx[:b1 - b0]
        np.multiply(step[b0:b1, None], j[None, :], out=ix)
        ix += first[b0:b1, None]
        blk = np.take(flat, ix)
        rv = rev[b0:b1]
        blk[rv] = 3 - blk[rv]
        r[b0:b1] = blk
    if err > 0:
        ne = rng.binomial(r.size, err)
        at = rng.integers(0, r.size, ne)
        rf = r.reshape(-1)
        rf[at] = (rf[at] + rng.integers(1, 4, ne, dtype=np.uint8)) & 3
    return r


def k12_genomes():
    """the three small genomes of tests/golden/make_golden_k12.py (k - drlevel = 9), as FASTA texts by file name"""
    rng = np.random.default_rng(20260312)
    a = rng.integers(0, 4, 1_500_000, dtype=np.uint8)
    b = a.copy()
    hit = rng.random(len(b)) < 0.002
    b[hit] = (b[hit] + rng.integers(1, 4, int(hit.sum()), dtype=np.uint8)) & 3
    c = rng.integers(0, 4, 600_000, dtype=np.uint8)
    nm = np.zeros(len(a), dtype=bool)
    nm[rng.integers(0, len(a), 40)] = True
    return {"a.fa": fasta_text(a, b"a random genome", n_mask=nm), "b.fa": fasta_text(b, b"a with 0.2 % substitutions"),
            "c.fa": fasta_text(c, b"unrelated")}


def k12_mode_inputs():
    """inputs of tests/golden/make_golden_k12_modes.py (k - drlevel = 9 with -u, fastq -n 2, -A): a 1.2 Mb genome whose first
    300 kb come twice (tuples -u drops), and 30 000 reads of 150 bp from a 600 kb genome (7.5 x: occurrence counts that matter)"""
    rng = np.random.default_rng(20260412)
    g = rng.integers(0, 4, 900_000, dtype=np.uint8)
    dup = np.concatenate([g, g[:300_000]])
    src = rng.integers(0, 4, 600_000, dtype=np.uint8)
    reads = []
    for _ in range(30_000):
        s = int(rng.integers(0, len(src) - 150))
        r = src[s:s + 150].copy()
        if rng.random() < 0.5:
            r = (3 - r)[::-1]
        reads.append(r)
    return {"dup.fa": fasta_text(dup, b"first 300 kb twice"), "reads.fq": fastq_text(reads)}
