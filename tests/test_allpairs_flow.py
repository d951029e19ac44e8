"""`kssd dist --allpairs`: stage I, the exchange and the search in ONE command, sketches resident on the devices
(kssd_gpu_resident_*, csrc/kssd_resident.inc).  The host orchestration (which device sketches and owns which inputs,
kssd_shard_plan) runs without a GPU; the flow itself is -m gpu and must leave what the two-command flow leaves and what the
REAL reference binary left for the same genomes (tests/golden/make_golden_allpairs.py)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import public_kssd_amd as K

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
BIN = os.path.join(ROOT, "public_kssd_amd", "kssd")


def plan(devices, n_files):
    L = K.host_lib()
    L.kssd_shard_plan.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_void_p]
    dv = np.asarray(devices, dtype=np.int32)
    first = np.zeros(len(dv) + 1, dtype=np.uint32)
    rc = L.kssd_shard_plan(dv.ctypes.data, len(dv), n_files, first.ctypes.data)
    return rc, first.tolist()


def test_inputs_are_dealt_out_in_equal_contiguous_runs():
    """the layout an all-gather of fixed-size units leaves: every device but the last ones holds ceil(n / devices) inputs, so a
    genome's number on every device is its input's index"""
    assert plan([0], 7) == (0, [0, 7])
    assert plan([0, 1, 2, 3], 10) == (0, [0, 3, 6, 9, 10])
    assert plan([3, 1], 9) == (0, [0, 5, 9])
    assert plan([0, 1, 2, 3], 9) == (0, [0, 3, 6, 9, 9])       # the last device: nothing
    assert plan([0, 1, 2], 2) == (0, [0, 1, 2, 2])
    assert plan([5, 6], 0) == (0, [0, 0, 0])
    for n_dev in range(1, 9):
        for n in (0, 1, 7, 8, 9, 1000, 1001):
            rc, first = plan(list(range(n_dev)), n)
            per = -(-n // n_dev) if n else 0
            assert rc == 0 and first[0] == 0 and first[-1] == n
            sizes = np.diff(first)
            assert all(s == per for s in sizes[:max(0, int(np.count_nonzero(sizes)) - 1)]) and sizes.max(initial=0) <= per


def test_a_device_named_twice_is_refused():
    assert plan([0, 0], 10)[0] == -102                          # KSSD_HOST_ERR_PARAM: one rank per device
    assert plan([1, 2, 1], 10)[0] == -102
    assert plan([0, -1], 10)[0] == -102
    assert plan([], 10)[0] == -102


def test_command_refuses_a_device_list_with_a_repeat_before_it_touches_a_gpu(tmp_path):
    """KSSD_DEVICE_LIST=0,0 kssd dist --allpairs: refused with the reason and EINVAL -- by the plan, i.e. also on a machine
    without any device (which is where this test runs in the CPU suite)"""
    d = str(tmp_path)
    K.Shuf.generate(10, 6, 3, seed=3).write(os.path.join(d, "s.shuf"))
    r = subprocess.run([BIN, "dist", "-L", "s.shuf", "-o", "out", "--allpairs", os.path.join(G, "ref_fa")], cwd=d, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=120, env=dict(os.environ, KSSD_DEVICE_LIST="0,0"))
    assert r.returncode == 22 and b"names a device twice" in r.stdout, r.stdout.decode()
    assert not os.path.exists(os.path.join(d, "out", "cofiles.stat"))
    r = subprocess.run([BIN, "dist", "-L", "s.shuf", "-o", "out", "--allpairs", "--byread", os.path.join(G, "ref_fa")], cwd=d,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)
    assert r.returncode != 0 and b"--allpairs with --byread" in r.stdout


def test_command_says_what_allpairs_cannot_do_before_it_sketches_anything(tmp_path):
    """--allpairs extends the stage I branch of dist_dispatch (command_dist.c:159-189): with -r, with a sketch directory as its
    input, or onto an existing sharedk_ct.dat (the reference refuses to overwrite it, :707-748) the command stops with the reason
    before stage I has run or a file of the output directory has been rewritten; a device list that is not a list of numbers is
    refused in every mode"""
    d = str(tmp_path)
    K.Shuf.generate(10, 6, 3, seed=3).write(os.path.join(d, "s.shuf"))
    fa = os.path.join(G, "ref_fa")
    run = lambda args, env=None: subprocess.run([BIN] + args, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120,
                                                env=dict(os.environ, **(env or {})))
    r = run(["dist", "-L", "s.shuf", "-o", "out", "--allpairs", "-r", fa, fa])
    assert r.returncode == 22 and b"--allpairs: all-pairs among the inputs of this run" in r.stdout, r.stdout.decode()
    os.makedirs(os.path.join(d, "sk"))
    open(os.path.join(d, "sk", "cofiles.stat"), "wb").write(b"\0" * 32)                           # what dist_dispatch probes for (command_dist.c:62-63)
    r = run(["dist", "-L", "s.shuf", "-o", "out", "--allpairs", "sk"])
    assert r.returncode == 22 and b"holds sketches" in r.stdout, r.stdout.decode()
    os.makedirs(os.path.join(d, "full"))
    open(os.path.join(d, "full", "sharedk_ct.dat"), "wb").write(b"1234")
    r = run(["dist", "-L", "s.shuf", "-o", "full", "--allpairs", fa])
    assert r.returncode == 17 and b"mco_cbdco_nobin_dist()" in r.stdout, r.stdout.decode()        # EEXIST, before stage I
    assert os.listdir(os.path.join(d, "full")) == ["sharedk_ct.dat"]
    r = run(["dist", "-L", "s.shuf", "-o", "out", "--allpairs"])
    assert r.returncode == 22 and b"no input sequences" in r.stdout
    for bad in ("0,a", "-1", "0,,1", "1,", "x", ",".join(str(i) for i in range(65))):
        r = run(["dist", "-L", "s.shuf", "-o", "out", fa], env={"KSSD_DEVICE_LIST": bad})
        assert r.returncode == 22 and b"KSSD_DEVICE_LIST" in r.stdout, (bad, r.stdout.decode())
    assert not os.path.exists(os.path.join(d, "out", "cofiles.stat"))


def _run(args, cwd, env=None):
    r = subprocess.run([BIN] + [str(a) for a in args], cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900,
                       env=dict(os.environ, **env) if env else None)
    assert r.returncode == 0, r.stdout.decode() + r.stderr.decode()
    return r.stdout.decode(), r.stderr.decode()


@pytest.mark.gpu
def test_one_command_all_pairs_equals_the_two_command_flow_and_the_reference(tmp_path):
    """--gpus 1: the flow goes through the exchange code with a one-rank RCCL communicator (KSSD_TIMING shows the stage), and
    combco.0 / sharedk_ct.dat / distance.out are byte for byte what `kssd dist -L ..; kssd dist -r ..` leave; against the
    reference binary's goldens the matrix is compared by file name and the report as a set of lines (its input order is
    shuffled by the clock, ours is sorted)"""
    import json
    meta = json.load(open(os.path.join(G, "golden.json")))
    d = str(tmp_path)
    fa = os.path.join(G, "ref_fa")
    K.Shuf.generate(10, 6, 3, seed=meta["seed"]).write(os.path.join(d, "L3K10.shuf"))
    out, err = _run(["dist", "-L", "L3K10.shuf", "-o", "one", "--allpairs", "--keepskf", "--gpus", 1, fa], d,
                    env={"KSSD_TIMING": "1", "KSSD_EXCHANGE_ONE_RANK": "1"})     # through RCCL although there is nobody to exchange with
    assert '"kssd_timing": "allpairs"' in err and '"gpus": 1' in err
    _run(["dist", "-L", "L3K10.shuf", "-o", "one_local", "--allpairs", "--keepskf", fa], d)   # the default on one device: unpacked in place
    for f in ("sharedk_ct.dat", "distance.out"):
        assert open(os.path.join(d, "one", f), "rb").read() == open(os.path.join(d, "one_local", f), "rb").read(), f
    _run(["dist", "-L", "L3K10.shuf", "-o", "two", fa], d)
    _run(["dist", "-r", "two", "-o", "two_out", "--keepskf", "two"], d)
    for f in ("combco.0", "combco.index.0", "cofiles.stat"):
        assert open(os.path.join(d, "one", f), "rb").read() == open(os.path.join(d, "two", f), "rb").read(), f
    assert open(os.path.join(d, "one", "sharedk_ct.dat"), "rb").read() == open(os.path.join(d, "two_out", "sharedk_ct.dat"), "rb").read()
    assert open(os.path.join(d, "one", "distance.out"), "rb").read() == open(os.path.join(d, "two_out", "distance.out"), "rb").read()
    # the reference binary's all-pairs of the same genomes
    W = np.load(os.path.join(G, "allpairs.npz"))
    names = sorted(os.listdir(fa))
    sh = np.fromfile(os.path.join(d, "one", "sharedk_ct.dat"), np.uint32).reshape(len(names), len(names))
    qi = [names.index(str(n)) for n in W["qry_names"]]
    ri = [names.index(str(n)) for n in W["ref_names"]]
    assert np.array_equal(sh[np.ix_(qi, ri)], W["shared"])
    got = open(os.path.join(d, "one", "distance.out")).read().replace(fa, "FA")
    want = W["distance_M0_O2"].tobytes().decode()
    assert got.splitlines()[0] == want.splitlines()[0] and sorted(got.splitlines()[1:]) == sorted(want.splitlines()[1:])
    # report options reach the one-command flow too (-M 1 -N 3), and the count file is not kept without --keepskf
    _run(["dist", "-L", "L3K10.shuf", "-o", "opt", "--allpairs", "-M", 1, "-N", 3, fa], d)
    assert not os.path.exists(os.path.join(d, "opt", "sharedk_ct.dat"))
    got = open(os.path.join(d, "opt", "distance.out")).read().replace(fa, "FA")
    want = W["distance_M1_N3"].tobytes().decode()
    assert got.splitlines()[0] == want.splitlines()[0] and sorted(got.splitlines()[1:]) == sorted(want.splitlines()[1:])
    # -u: the keep rule is replayed on the host, the kept ids go back to the device (kssd_gpu_resident_put_host)
    _run(["dist", "-L", "L3K10.shuf", "-o", "u1", "--allpairs", "--keepskf", "-u", fa], d)
    _run(["dist", "-L", "L3K10.shuf", "-o", "u2", "-u", fa], d)
    _run(["dist", "-r", "u2", "-o", "u2_out", "--keepskf", "u2"], d)
    assert open(os.path.join(d, "u1", "sharedk_ct.dat"), "rb").read() == open(os.path.join(d, "u2_out", "sharedk_ct.dat"), "rb").read()
    assert open(os.path.join(d, "u1", "distance.out"), "rb").read() == open(os.path.join(d, "u2_out", "distance.out"), "rb").read()


@pytest.mark.gpu
def test_resident_sets_through_the_c_abi(shuf_l3k10):
    """kssd_gpu_resident_*: two sketch contexts of one device put their batches into one set (slots out of order), the
    all-pairs matrix and planes equal kssd_gpu_dist on the same CSR; a device named twice and a slot never put are refused"""
    import kssd_oracle as ko
    from synth import clade_genomes, fasta_text
    L = K.gpu_lib()
    vp = C.c_void_p
    L.kssd_gpu_resident_create.argtypes = [C.POINTER(vp), C.c_int, C.c_uint32]
    L.kssd_gpu_resident_destroy.argtypes = [vp]
    L.kssd_gpu_resident_destroy.restype = None
    L.kssd_gpu_resident_put.argtypes = [vp, vp, C.c_uint32, C.c_uint32]
    L.kssd_gpu_resident_put_host.argtypes = [vp, C.c_uint32, C.c_uint32, vp, vp]
    L.kssd_gpu_resident_sizes.argtypes = [vp, vp]
    L.kssd_gpu_resident_allpairs.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp, vp, vp]
    texts = [fasta_text(c, nm, n_mask=m) for nm, c, m in clade_genomes(3, 4, 120_000, seed=11)]     # 12 genomes
    a, b = K.GpuCtx(shuf_l3k10, 0), K.GpuCtx(shuf_l3k10, 0)
    r = vp()
    assert L.kssd_gpu_resident_create(C.byref(r), 0, 12) == 0
    try:
        offs, idss = {}, {}
        for ctx, first, n in ((a, 5, 4), (b, 0, 5), (a, 9, 3)):                                      # slots 5..8, 0..4, 9..11
            off, ids = ctx.sketch_fasta_texts(texts[first:first + n])
            offs[first], idss[first] = off, ids
            if first == 9:    # the last one from host arrays
                assert L.kssd_gpu_resident_put_host(r, first, n, off.ctypes.data, ids.ctypes.data) == 0
            else:
                assert L.kssd_gpu_resident_put(r, ctx.h, first, n) == 0
        assert L.kssd_gpu_resident_put(r, a.h, 0, 3) != 0                                            # a slot is put once
        sizes = np.zeros(12, np.uint32)
        assert L.kssd_gpu_resident_sizes(r, sizes.ctypes.data) == 0
        off_all = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
        ids_all = np.concatenate([idss[0], idss[5], idss[9]])
        for g in range(12):
            assert np.array_equal(np.sort(ids_all[int(off_all[g]):int(off_all[g + 1])]), np.sort(ko.Sketcher(shuf_l3k10.table, 10, 6, 3).fasta(texts[g])))
        shared = np.zeros((12, 12), np.uint32)
        planes = [np.zeros((12, 12), np.float64) for _ in range(4)]
        sets = (vp * 1)(r)
        assert L.kssd_gpu_resident_allpairs(sets, 1, 20, shared.ctypes.data, *[p.ctypes.data for p in planes]) == 0
        want = a.dist(off_all, ids_all, off_all, ids_all)
        assert np.array_equal(shared, want[0]) and shared.trace() == sizes.sum()
        for p, w in zip(planes, want[1:]):
            assert np.array_equal(p.view(np.int64), np.asarray(w).view(np.int64))
        two = (vp * 2)(r, r)                                                                          # one rank per device
        assert L.kssd_gpu_resident_allpairs(two, 2, 20, shared.ctypes.data, None, None, None, None) == K.capi.ERR_PARAM
        r2 = vp()
        assert L.kssd_gpu_resident_create(C.byref(r2), 0, 3) == 0
        assert L.kssd_gpu_resident_put(r2, a.h, 0, 3) == 0                                           # slots 0..2 only ... (a's last call: 3 genomes)
        L.kssd_gpu_resident_destroy(r2)
        r3 = vp()
        assert L.kssd_gpu_resident_create(C.byref(r3), 0, 5) == 0
        assert L.kssd_gpu_resident_put(r3, a.h, 0, 3) == 0                                           # ... of five: slots 3, 4 never put
        assert L.kssd_gpu_resident_allpairs((vp * 1)(r3), 1, 20, shared.ctypes.data, None, None, None, None) == K.capi.ERR_PARAM
        L.kssd_gpu_resident_destroy(r3)
    finally:
        L.kssd_gpu_resident_destroy(r)
        a.close()
        b.close()


@pytest.mark.gpu
def test_n_rank_orchestration_on_one_device_leaves_the_one_device_files(tmp_path):
    """KSSD_EXCHANGE_FAKE_RANKS=n: `kssd dist --allpairs` plans, sketches, keeps, exchanges and searches as n ranks -- every rank's
    buffers on device 0, the collective replaced by device-to-device copies of the bytes an all-gather delivers -- and must leave
    byte for byte what --gpus 1 leaves: unit padding, a short and an EMPTY last set, the global numbering base[d] * N, one host
    thread per rank writing its rows into one mapping, in both partitions of the matrix (own index + transposed write, the
    default; the full index, KSSD_ALLPAIRS_FULL_INDEX=1)"""
    import json
    from synth import clade_genomes, fasta_text
    meta = json.load(open(os.path.join(G, "golden.json")))
    d = str(tmp_path)
    K.Shuf.generate(10, 6, 3, seed=meta["seed"]).write(os.path.join(d, "L3K10.shuf"))
    fa = os.path.join(d, "fa")
    os.makedirs(fa)
    for i, (nm, codes, m) in enumerate(clade_genomes(3, 4, 150_000, seed=23)[:11]):          # 11 genomes: 4 ranks hold 3, 3, 3, 2
        open(os.path.join(fa, "g%02d.fasta" % i), "wb").write(fasta_text(codes, nm, n_mask=m))
    open(os.path.join(fa, "g11_empty.fasta"), "wb").write(b">nothing\nACGTNNNN\n")                 # a sketch without ids among them
    _run(["dist", "-L", "L3K10.shuf", "-o", "one", "--allpairs", "--keepskf", fa], d)
    want_sk = open(os.path.join(d, "one", "sharedk_ct.dat"), "rb").read()
    want_txt = open(os.path.join(d, "one", "distance.out"), "rb").read()
    assert len(want_sk) == 12 * 12 * 4
    for n, full in ((2, False), (4, False), (5, False), (7, False), (13, False), (4, True), (5, True)):  # 13 ranks: more ranks than genomes
        out = "r%d%s" % (n, "f" if full else "")
        env = {"KSSD_EXCHANGE_FAKE_RANKS": str(n), "KSSD_TIMING": "1"}
        if full:
            env["KSSD_ALLPAIRS_FULL_INDEX"] = "1"
        o, err = _run(["dist", "-L", "L3K10.shuf", "-o", out, "--allpairs", "--keepskf", fa], d, env=env)
        line = [json.loads(l) for l in err.splitlines() if '"resident_allpairs"' in l][0]
        assert line["ranks"] == n and "device-to-device" in line["exchange"], line
        assert ("full index" in line["partition"]) == full, line
        assert open(os.path.join(d, out, "sharedk_ct.dat"), "rb").read() == want_sk, (n, full)
        assert open(os.path.join(d, out, "distance.out"), "rb").read() == want_txt, (n, full)
        for f in ("combco.0", "combco.index.0", "cofiles.stat"):
            assert open(os.path.join(d, out, f), "rb").read() == open(os.path.join(d, "one", f), "rb").read(), (n, f)


@pytest.mark.gpu
def test_resident_sets_as_several_ranks_through_the_c_abi(shuf_l3k10):
    """kssd_gpu_resident_allpairs over three sets (5 + 5 + 2 genomes, KSSD_EXCHANGE_FAKE_RANKS: all on device 0) with all four
    planes: the matrix and every plane's bits equal kssd_gpu_dist on the same CSR -- the transposing metrics kernel computes a
    pair exactly as the rows kernel's epilogue does"""
    from synth import clade_genomes, fasta_text
    L = K.gpu_lib()
    vp = C.c_void_p
    L.kssd_gpu_resident_create.argtypes = [C.POINTER(vp), C.c_int, C.c_uint32]
    L.kssd_gpu_resident_destroy.argtypes = [vp]
    L.kssd_gpu_resident_destroy.restype = None
    L.kssd_gpu_resident_put.argtypes = [vp, vp, C.c_uint32, C.c_uint32]
    L.kssd_gpu_resident_allpairs.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp, vp, vp]
    texts = [fasta_text(c, nm, n_mask=m) for nm, c, m in clade_genomes(3, 4, 120_000, seed=12)]     # 12 genomes
    ctx = K.GpuCtx(shuf_l3k10, 0)
    sets = []
    os.environ["KSSD_EXCHANGE_FAKE_RANKS"] = "3"
    try:
        offs, idl = [np.zeros(1, np.uint64)], []
        for first, n in ((0, 5), (5, 5), (10, 2)):
            r = vp()
            assert L.kssd_gpu_resident_create(C.byref(r), 0, n) == 0
            sets.append(r)
            off, ids = ctx.sketch_fasta_texts(texts[first:first + n])
            assert L.kssd_gpu_resident_put(r, ctx.h, 0, n) == 0
            offs.append(off[1:] + offs[-1][-1])
            idl.append(ids)
        off_all, ids_all = np.concatenate(offs).astype(np.uint64), np.concatenate(idl)
        want = ctx.dist(off_all, ids_all, off_all, ids_all)
        for full in (False, True):
            if full:
                os.environ["KSSD_ALLPAIRS_FULL_INDEX"] = "1"
            shared = np.full((12, 12), 77, np.uint32)
            planes = [np.full((12, 12), 7.0, np.float64) for _ in range(4)]
            assert L.kssd_gpu_resident_allpairs((vp * 3)(*sets), 3, 20, shared.ctypes.data, *[p.ctypes.data for p in planes]) == 0
            assert np.array_equal(shared, want[0]) and np.array_equal(shared, shared.T)
            for p, w in zip(planes, want[1:]):
                assert np.array_equal(p.view(np.int64), np.asarray(w).view(np.int64)), full
    finally:
        os.environ.pop("KSSD_EXCHANGE_FAKE_RANKS", None)
        os.environ.pop("KSSD_ALLPAIRS_FULL_INDEX", None)
        for r in sets:
            L.kssd_gpu_resident_destroy(r)
        ctx.close()
