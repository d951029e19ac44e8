"""ctypes binding of oracle/libkssd_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product (public_kssd_amd/) never does.  See oracle/kssd_oracle.h for what each call restates.
Also holds numpy readers for the reference's on-disk formats and a runner for
the real reference binary oracle/_ref/kssd.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libkssd_oracle.so")
REF_BIN = os.path.join(HERE, "_ref", "kssd")

ERRORS = {-2: "capacity", -3: "header", -4: "empty", -5: "param", -6: "io", -7: "bufsz"}


class OracleError(RuntimeError):
    def __init__(self, code):
        super().__init__("kssd oracle error %d (%s)" % (code, ERRORS.get(code, "?")))
        self.code = code


class Params(C.Structure):
    _fields_ = [("shuf_id", C.c_int), ("k", C.c_int), ("subk", C.c_int), ("drlevel", C.c_int),
                ("TL", C.c_int), ("out", C.c_int), ("comp_num", C.c_int), ("comp_bits", C.c_int),
                ("rc_shift", C.c_int), ("tupmask", C.c_uint64), ("domask", C.c_uint64),
                ("undomask", C.c_uint64), ("dim_end", C.c_int64), ("hashsize", C.c_uint32),
                ("hashlimit", C.c_uint32)]


class Metric(C.Structure):
    _fields_ = [("metric", C.c_double), ("dist", C.c_double), ("pv", C.c_double), ("fdr", C.c_double),
                ("ci_m1", C.c_double), ("ci_m2", C.c_double), ("ci_d1", C.c_double), ("ci_d2", C.c_double),
                ("rs_u", C.c_uint32), ("skipped", C.c_int)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            subprocess.check_call(["make", "-C", HERE, LIB_PATH])
        L = C.CDLL(LIB_PATH)
        L.ko_open.restype = C.c_void_p
        L.ko_open.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.ko_close.argtypes = [C.c_void_p]
        L.ko_get_params.restype = C.POINTER(Params)
        L.ko_get_params.argtypes = [C.c_void_p]
        L.ko_params_init.argtypes = [C.POINTER(Params), C.c_int, C.c_int, C.c_int, C.c_int]
        L.ko_fasta2co.restype = C.c_long
        L.ko_fasta2co.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]
        L.ko_fastq2koc.restype = C.c_long
        L.ko_fastq2koc.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        L.ko_reads2mco.restype = C.c_long
        L.ko_reads2mco.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                   C.POINTER(C.c_uint64)]
        L.ko_fastq2co.restype = C.c_long
        L.ko_fastq2co.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                  C.c_size_t]
        L.ko_sketch_file.restype = C.c_long
        L.ko_sketch_file.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                     C.c_void_p, C.c_size_t]
        L.ko_sketch_files.restype = C.c_long
        L.ko_sketch_files.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_char_p), C.c_int,
                                      C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]
        L.ko_sketch_texts.restype = C.c_long
        L.ko_sketch_texts.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_char_p),
                                      C.POINTER(C.c_size_t), C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]
        L.ko_shared_counts.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                       C.c_int]
        L.ko_build_index.restype = C.c_long
        L.ko_build_index.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ko_output_ctrl.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_double, C.c_uint64, C.POINTER(Metric)]
        L.ko_format_line.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_char_p, C.c_uint32, C.c_uint32,
                                     C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_uint64]
        L.ko_dist_print.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_char_p,
                                    C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int]
        _lib = L
    return _lib


def params(k, subk, drlevel, shuf_id=0):
    p = Params()
    r = lib().ko_params_init(C.byref(p), shuf_id, k, subk, drlevel)
    if r:
        raise OracleError(r)
    return p


class Sketcher:
    """One reference-style sketching context (one hash table), bound to a .shuf permutation."""

    def __init__(self, table, k, subk, drlevel, shuf_id=0):
        self.table = np.ascontiguousarray(table, dtype=np.int32)
        assert self.table.size == 16 ** subk
        self.h = lib().ko_open(self.table.ctypes.data, shuf_id, k, subk, drlevel)
        if not self.h:
            raise OracleError(-5)
        self.p = lib().ko_get_params(self.h).contents
        self._cap = int(self.p.hashsize)
        self._ids = np.empty(self._cap, dtype=np.uint32)
        self._comps = np.empty(self._cap, dtype=np.uint8)

    def close(self):
        if self.h:
            lib().ko_close(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def _ret(self, n, with_comps):
        if n < 0:
            raise OracleError(n)
        if with_comps:
            return self._ids[:n].copy(), self._comps[:n].copy()
        return self._ids[:n].copy()

    def fasta(self, text, uniq=False, with_comps=False):
        """ids in the reference's file order (hash-slot order)"""
        text = bytes(text)
        n = lib().ko_fasta2co(self.h, text, len(text), int(uniq), self._ids.ctypes.data, self._comps.ctypes.data,
                              self._cap)
        return self._ret(n, with_comps)

    def fastq(self, text, Q=0, M=1, with_comps=False):
        text = bytes(text)
        n = lib().ko_fastq2co(self.h, text, len(text), Q, M, self._ids.ctypes.data, self._comps.ctypes.data, self._cap)
        return self._ret(n, with_comps)

    def fastq_koc(self, text):
        """dist -A: (ids, counts) in the reference's file order"""
        text = bytes(text)
        counts = np.zeros(self._cap, np.uint16)
        n = lib().ko_fastq2koc(self.h, text, len(text), self._ids.ctypes.data, self._comps.ctypes.data,
                               counts.ctypes.data, self._cap)
        if n < 0:
            raise OracleError(n)
        return self._ids[:n].copy(), counts[:n].copy()

    def byread(self, text):
        """dist --byread: (ids, comps, read_of, n_reads) -- the sampled k-mers in sequence order, repeats included"""
        text = bytes(text)
        read_of = np.zeros(self._cap, np.uint32)
        nr = C.c_uint64(0)
        n = lib().ko_reads2mco(self.h, text, len(text), self._ids.ctypes.data, self._comps.ctypes.data,
                               read_of.ctypes.data, self._cap, C.byref(nr))
        if n < 0:
            raise OracleError(n)
        return self._ids[:n].copy(), self._comps[:n].copy(), read_of[:n].copy(), int(nr.value)

    def byread_files(self, text):
        """what reads2mco leaves on disk: {component: (combco bytes as u32[], combco.index as i64[])}"""
        ids, comps, read_of, nr = self.byread(text)
        out = {}
        for c in range(self.p.comp_num):
            sel = comps == c
            cnt = np.bincount(read_of[sel], minlength=nr + 1).astype(np.int64)
            out[c] = (ids[sel].copy(), np.cumsum(cnt))
        return out

    def file(self, path, is_fastq=False, uniq=False, Q=0, M=1):
        n = lib().ko_sketch_file(self.h, os.fsencode(path), int(is_fastq), int(uniq), Q, M, self._ids.ctypes.data,
                                 self._comps.ctypes.data, self._cap)
        return self._ret(n, False)


def sketch_texts(table, k, subk, drlevel, texts, threads=1):
    """CSR (off, ids) for a list of FASTA byte strings, OpenMP over texts."""
    table = np.ascontiguousarray(table, dtype=np.int32)
    n = len(texts)
    arr = (C.c_char_p * n)(*texts)
    lens = (C.c_size_t * n)(*[len(t) for t in texts])
    cap = sum(len(t) for t in texts) // 64 + 4096 * n
    off = np.zeros(n + 1, dtype=np.uint64)
    ids = np.empty(cap, dtype=np.uint32)
    r = lib().ko_sketch_texts(table.ctypes.data, 0, k, subk, drlevel, arr, lens, n, threads, off.ctypes.data,
                              ids.ctypes.data, cap)
    if r < 0:
        raise OracleError(r)
    return off, ids[:r].copy()


def sketch_files(table, k, subk, drlevel, paths, threads=1):
    table = np.ascontiguousarray(table, dtype=np.int32)
    n = len(paths)
    arr = (C.c_char_p * n)(*[os.fsencode(p) for p in paths])
    cap = 2 * 1258285 if n < 64 else n * 40000
    off = np.zeros(n + 1, dtype=np.uint64)
    ids = np.empty(cap, dtype=np.uint32)
    r = lib().ko_sketch_files(table.ctypes.data, 0, k, subk, drlevel, arr, n, threads, off.ctypes.data,
                              ids.ctypes.data, cap)
    if r < 0:
        raise OracleError(r)
    return off, ids[:r].copy()


def shared_counts(roff, rids, qoff, qids, threads=1):
    roff = np.ascontiguousarray(roff, dtype=np.uint64)
    qoff = np.ascontiguousarray(qoff, dtype=np.uint64)
    rids = np.ascontiguousarray(rids, dtype=np.uint32)
    qids = np.ascontiguousarray(qids, dtype=np.uint32)
    R, Q = len(roff) - 1, len(qoff) - 1
    out = np.zeros((Q, R), dtype=np.uint32)
    lib().ko_shared_counts(roff.ctypes.data, rids.ctypes.data, R, qoff.ctypes.data, qids.ctypes.data, Q,
                           out.ctypes.data, threads)
    return out


def build_index(roff, rids):
    roff = np.ascontiguousarray(roff, dtype=np.uint64)
    rids = np.ascontiguousarray(rids, dtype=np.uint32)
    n = int(roff[-1])
    uid = np.empty(max(n, 1), dtype=np.uint32)
    upos = np.empty(n + 1, dtype=np.uint64)
    post = np.empty(max(n, 1), dtype=np.uint32)
    U = lib().ko_build_index(roff.ctypes.data, rids.ctypes.data, len(roff) - 1, uid.ctypes.data, upos.ctypes.data,
                             post.ctypes.data)
    return uid[:U].copy(), upos[:U + 1].copy(), post[:n].copy()


def output_ctrl(X, Y, s, kmerlen, dim_rd_len, metric_sel=0, correction=0, dthreshold=1.0, cmprsn_num=1):
    m = Metric()
    lib().ko_output_ctrl(X, Y, s, kmerlen, dim_rd_len, metric_sel, correction, dthreshold, cmprsn_num, C.byref(m))
    return m


def metrics_batch(X, Y, S, kmerlen):
    """(J, MashD, C, AafD) f64 arrays for broadcastable uint32 arrays, rs = 0, one C loop over the host libm"""
    X, Y, S = np.broadcast_arrays(np.asarray(X, np.uint32), np.asarray(Y, np.uint32), np.asarray(S, np.uint32))
    shape = S.shape
    X, Y, S = (np.ascontiguousarray(a).reshape(-1) for a in (X, Y, S))
    out = [np.empty(S.size) for _ in range(4)]
    f = lib().ko_metrics_batch
    f.restype = None
    f.argtypes = [C.c_void_p] * 3 + [C.c_size_t, C.c_int] + [C.c_void_p] * 4
    f(X.ctypes.data, Y.ctypes.data, S.ctypes.data, S.size, kmerlen, *[o.ctypes.data for o in out])
    return tuple(o.reshape(shape) for o in out)


def metrics_arrays(X, Y, S, kmerlen):
    """(J, MashD, C, AafD) f64 arrays for broadcastable uint32 arrays, rs=0 (no correction)."""
    X, Y, S = np.broadcast_arrays(np.asarray(X, np.uint32), np.asarray(Y, np.uint32), np.asarray(S, np.uint32))
    J = np.empty(S.shape)
    MD = np.empty(S.shape)
    Cc = np.empty(S.shape)
    AD = np.empty(S.shape)
    it = np.nditer([X, Y, S], flags=["multi_index"])
    for x, y, s in it:
        a = output_ctrl(int(x), int(y), int(s), kmerlen, 0, 0)
        b = output_ctrl(int(x), int(y), int(s), kmerlen, 0, 1)
        J[it.multi_index], MD[it.multi_index] = a.metric, a.dist
        Cc[it.multi_index], AD[it.multi_index] = b.metric, b.dist
    return J, MD, Cc, AD


def format_line(qname, rname, X, Y, s, kmerlen, dim_rd_len, metric_sel=0, pfield=2, correction=0, dthreshold=1.0,
                cmprsn_num=1):
    buf = C.create_string_buffer(2048)
    n = lib().ko_format_line(buf, 2048, os.fsencode(qname), os.fsencode(rname), X, Y, s, kmerlen, dim_rd_len,
                             metric_sel, pfield, correction, dthreshold, cmprsn_num)
    return buf.raw[:n]


def dist_print(path, shared, ref_sz, qry_sz, refnames, qrynames, kmerlen, dim_rd_len, metric_sel=0, pfield=2,
               correction=0, dthreshold=1.0, n_max=0):
    shared = np.ascontiguousarray(shared, dtype=np.uint32)
    Q, R = shared.shape
    ref_sz = np.ascontiguousarray(ref_sz, dtype=np.uint32)
    qry_sz = np.ascontiguousarray(qry_sz, dtype=np.uint32)
    rn = b"".join(os.fsencode(n).ljust(256, b"\0") for n in refnames)
    qn = b"".join(os.fsencode(n).ljust(256, b"\0") for n in qrynames)
    r = lib().ko_dist_print(os.fsencode(path), shared.ctypes.data, R, Q, ref_sz.ctypes.data, qry_sz.ctypes.data, rn,
                            qn, kmerlen, dim_rd_len, metric_sel, pfield, correction, dthreshold, n_max)
    if r:
        raise OracleError(r)


# --------------------------------------------------------------------------------------------------
# reference on-disk formats -- numpy readers used by the tests
# --------------------------------------------------------------------------------------------------
def read_shuf(path):
    """(hdr dict, int32 table) -- command_shuffle.c:192-207"""
    with open(path, "rb") as f:
        hdr = np.frombuffer(f.read(16), dtype=np.int32)
        table = np.frombuffer(f.read(), dtype=np.int32)
    h = dict(id=int(hdr[0]), k=int(hdr[1]), subk=int(hdr[2]), drlevel=int(hdr[3]))
    assert table.size == 16 ** h["subk"], (table.size, h)
    return h, table


def read_stat(path, mco=False):
    """cofiles.stat (co_dstat_t, 32 B, global_basic.h:94-103) / mcofiles.stat (mco_dstat_t, 20 B,
    command_dist.h:57-64) + u32 sizes[n] + char names[n][256]"""
    raw = open(path, "rb").read()
    if mco:
        shuf_id, kmerlen, dim_rd_len, comp_num, n = np.frombuffer(raw[:20], dtype=np.int32)
        hdr = dict(shuf_id=int(np.uint32(shuf_id)), kmerlen=int(kmerlen), dim_rd_len=int(dim_rd_len),
                   comp_num=int(comp_num), infile_num=int(n))
        o = 20
    else:
        shuf_id = int(np.frombuffer(raw[:4], dtype=np.uint32)[0])
        koc = raw[4]
        kmerlen, dim_rd_len, comp_num, n = np.frombuffer(raw[8:24], dtype=np.int32)
        all_ctx = int(np.frombuffer(raw[24:32], dtype=np.uint64)[0])
        hdr = dict(shuf_id=shuf_id, koc=int(koc), kmerlen=int(kmerlen), dim_rd_len=int(dim_rd_len),
                   comp_num=int(comp_num), infile_num=int(n), all_ctx_ct=all_ctx)
        o = 32
    n = hdr["infile_num"]
    sizes = np.frombuffer(raw[o:o + 4 * n], dtype=np.uint32).copy()
    o += 4 * n
    names = [raw[o + 256 * i:o + 256 * (i + 1)].split(b"\0")[0].decode() for i in range(n)]
    assert len(raw) == o + 256 * n, (len(raw), o, n)
    return hdr, sizes, names


def read_sketch_dir(d, comp=0):
    """(hdr, names, off uint64[n+1], ids uint32) from cofiles.stat + combco.<c> + combco.index.<c>"""
    hdr, sizes, names = read_stat(os.path.join(d, "cofiles.stat"))
    off = np.fromfile(os.path.join(d, "combco.index.%d" % comp), dtype=np.uint64)
    ids = np.fromfile(os.path.join(d, "combco.%d" % comp), dtype=np.uint32)
    assert off.size == hdr["infile_num"] + 1 and off[-1] == ids.size
    if hdr["comp_num"] == 1:
        assert np.array_equal(np.diff(off).astype(np.uint32), sizes)
    return hdr, names, off, ids


def sketch_sets_by_name(d):
    """{basename: sorted unique ids} -- the comparison key SURVEY.md section 4 prescribes"""
    hdr, names, off, ids = read_sketch_dir(d)
    return {os.path.basename(nm): np.sort(ids[int(off[i]):int(off[i + 1])]) for i, nm in enumerate(names)}


# --------------------------------------------------------------------------------------------------
# the real reference binary
# --------------------------------------------------------------------------------------------------
def have_ref():
    return os.access(REF_BIN, os.X_OK)


def run_ref(args, cwd=None, timeout=600, check=True):
    """Run oracle/_ref/kssd with argv[0]="kssd": the reference's main() writes one byte past a heap
    block when strlen(argv[0]) % 16 == 11 (kssd.c:29-30) and then crashes or hangs at random."""
    env = {k: v for k, v in os.environ.items() if k != "LD_PRELOAD"}  # (a sanitizer run of OUR host library must not instrument it)
    r = subprocess.run(["kssd"] + [str(a) for a in args], executable=REF_BIN, cwd=cwd, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=timeout, env=env)
    if check and r.returncode != 0:
        raise RuntimeError("reference kssd %s -> %d\n%s" % (args, r.returncode, r.stdout.decode(errors="replace")[-2000:]))
    return r
