"""ctypes bindings of the two in-tree native libraries.

  libkssd_gpu.so   HIP kernels behind the C ABI declared in include/kssd_gpu.h
  libkssd_host.so  host C (host/kssd_host.h): .shuf files, tokeniser / packer, on-disk formats

There is no Python or CPU implementation of the hot path in this package: if libkssd_gpu.so is not
built, or no gfx950 device is usable, the calls raise.
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GPU_LIB = os.environ.get("KSSD_GPU_LIB", os.path.join(HERE, "libkssd_gpu.so"))  # (profiles/: the -DKSSD_DEV build with its time stamps)
HOST_LIB = os.environ.get("KSSD_HOST_LIB", os.path.join(HERE, "libkssd_host.so"))  # (the sanitizer run points this at its own build)

CHUNK_BASES = 4096
CHUNK_WORDS = 256
CHUNK_MASKW = 128
SLACK_WORDS = 8

SKETCH_FASTA = 0
SKETCH_KEEP_ZERO = 1
SKETCH_UNIQ = 2
SKETCH_NO_CAPACITY = 4
SKETCH_FIRST_POS = 8
SKETCH_COUNTS = 16
SKETCH_BY_POS = 32
PHASE_PREP, PHASE_SCAN, PHASE_EXACT, PHASE_FINISH, PHASE_REPASS = 0, 1, 2, 3, 4

OK, ERR_HIP, ERR_PARAM, ERR_CAPACITY, ERR_OVERFLOW, ERR_UNSUPPORTED, ERR_NOMEM, ERR_NO_DEVICE, ERR_INPUT = 0, -1, -2, -3, -4, -5, -6, -7, -8

GPU_SYMBOLS = [
    "kssd_gpu_strerror", "kssd_gpu_last_hip_error", "kssd_gpu_create", "kssd_gpu_create_compact",
    "kssd_gpu_create_for_dist",
    "kssd_gpu_destroy", "kssd_gpu_get_info", "kssd_gpu_sketch_device", "kssd_gpu_sketch_status",
    "kssd_gpu_sketch_batch", "kssd_gpu_free", "kssd_gpu_index_build_device", "kssd_gpu_dist_device",
    "kssd_gpu_dist", "kssd_gpu_kernel_time", "kssd_gpu_set_kernel_timing", "kssd_gpu_scan_stats", "kssd_gpu_sketch_set_pos_output",
    "kssd_gpu_sketch_batch_pos", "kssd_gpu_set_union", "kssd_gpu_set_filter", "kssd_gpu_sketch_plan",
    "kssd_gpu_sketch_phase", "kssd_gpu_set_lds_sort_limit", "kssd_gpu_dist_multi", "kssd_gpu_device_count",
    "kssd_gpu_host_alloc", "kssd_gpu_host_free", "kssd_gpu_dist_select", "kssd_gpu_dist_device_long",
    "kssd_gpu_tokenise_fasta_device", "kssd_gpu_tokenise_status", "kssd_gpu_sketch_fasta_text",
    "kssd_gpu_tokenise_fastq_device", "kssd_gpu_tokenise_fastq_status", "kssd_gpu_sketch_fastq_text",
    "kssd_gpu_text_reserve", "kssd_gpu_text_put", "kssd_gpu_text_wait", "kssd_gpu_concat_units_device",
    "kssd_gpu_index_set_filter", "kssd_gpu_set_scan_grid", "kssd_gpu_warm_up", "kssd_gpu_set_fastq_quality", "kssd_gpu_set_fastq_reads", "kssd_gpu_allgather_sketches", "kssd_gpu_fasta_read_starts", "kssd_gpu_tuple_passes", "kssd_gpu_set_tuple_pass", "kssd_gpu_sketch_again",
    "kssd_gpu_index_status", "kssd_gpu_index_set_exact",
    "kssd_gpu_resident_create", "kssd_gpu_resident_destroy", "kssd_gpu_resident_put", "kssd_gpu_resident_put_host",
    "kssd_gpu_resident_sizes", "kssd_gpu_resident_allpairs", "kssd_gpu_runtime_path", "kssd_gpu_exchange_warm_up",
    "kssd_gpu_dist_device_transposed", "kssd_gpu_kernel_times", "kssd_gpu_dist_counts_device", "kssd_gpu_transpose_metrics_device",
    "kssd_gpu_host_register", "kssd_gpu_host_unregister", "kssd_gpu_mask_summarise_device", "kssd_gpu_sketch_set_mask_summary",
    "kssd_gpu_tokenise_fasta_device_summary",
]


class KssdError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("kssd error %d: %s" % (code, msg))
        self.code = code


class ShufHdr(C.Structure):
    _fields_ = [("id", C.c_int32), ("k", C.c_int32), ("subk", C.c_int32), ("drlevel", C.c_int32)]


class GpuInfo(C.Structure):
    _fields_ = [("k", C.c_int32), ("subk", C.c_int32), ("drlevel", C.c_int32), ("kmerlen", C.c_int32),
                ("dim_rd_len", C.c_int32), ("comp_num", C.c_int32), ("comp_bits", C.c_int32),
                ("dim_end", C.c_uint32), ("hashsize", C.c_uint32), ("hashlimit", C.c_uint32),
                ("device", C.c_int32), ("cu_count", C.c_int32)]


class _SketchSet(C.Structure):
    _fields_ = [("shuf_id", C.c_uint32), ("koc", C.c_int), ("kmerlen", C.c_int), ("dim_rd_len", C.c_int),
                ("comp_num", C.c_int), ("n", C.c_uint32), ("off", C.c_void_p), ("ids", C.c_void_p),
                ("names", C.c_void_p), ("counts", C.c_void_p), ("sub", C.c_void_p)]


class _PrintOpt(C.Structure):
    _fields_ = [("metric", C.c_int), ("pfield", C.c_int), ("correction", C.c_int), ("dthreshold", C.c_double),
                ("n_max", C.c_int), ("threads", C.c_int)]


class _Derived(C.Structure):
    _fields_ = [("k", C.c_int), ("subk", C.c_int), ("drlevel", C.c_int), ("kmerlen", C.c_int), ("dim_rd_len", C.c_int),
                ("comp_num", C.c_int), ("comp_bits", C.c_int), ("hashsize", C.c_uint32), ("hashlimit", C.c_uint32)]


class _Shuf(C.Structure):
    _fields_ = [("id", C.c_int32), ("k", C.c_int32), ("subk", C.c_int32), ("drlevel", C.c_int32),
                ("table", C.POINTER(C.c_int32))]


_gpu = None
_host = None
DEFAULT_LDS_SORT_LIMIT = 0  # tests: kssd_gpu_set_lds_sort_limit for every new GpuCtx (0 = the library's default)


def kernel_source_sha():
    """sha256 over the device sources (csrc/*, sorted by name): what recorded counter figures are stamped with
    (profiles/pmc_refresh.py writes it, bench.py compares) -- the GPU box has no git history to ask"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(HERE, "csrc")
    for nm in sorted(os.listdir(d)):
        if nm.endswith((".hip", ".inc", ".h")):
            h.update(nm.encode())
            h.update(open(os.path.join(d, nm), "rb").read())
    return h.hexdigest()


def runtime_paths():
    """which HIP runtime / RCCL / HSA files this process has mapped: {"hip": [...], "rccl": [...], "hsa": [...]}"""
    out = {"hip": set(), "rccl": set(), "hsa": set()}
    try:
        for ln in open("/proc/self/maps"):
            path = ln.split()[-1] if "/" in ln else ""
            base = os.path.basename(path)
            if base.startswith("libamdhip64.so"):
                out["hip"].add(path)
            elif base.startswith("librccl.so"):
                out["rccl"].add(path)
            elif base.startswith("libhsa-runtime64.so"):
                out["hsa"].add(path)
    except OSError:
        pass
    return {k: sorted(v) for k, v in out.items()}


def assert_single_runtime():
    """One process, one HIP runtime: raw hipStream_t handles, device pointers and RCCL communicators cross between torch and
    libkssd_gpu.so in the harness (bench.py, the tests), which only means something inside ONE runtime.  torch's libraries ask
    for "libamdhip64.so" by that name, so a runtime that came in earlier under its soname (libamdhip64.so.7, what
    libkssd_gpu.so asks for) does not satisfy them and a second one gets mapped; the other way round the loader hands
    libkssd_gpu.so the copy torch has brought.  Hence gpu_lib() imports torch first where it exists."""
    rp = runtime_paths()
    for k in ("hip", "rccl"):
        if len(rp[k]) > 1:
            raise RuntimeError("two %s libraries are mapped into this process: %s -- import torch before public_kssd_amd loads "
                               "libkssd_gpu.so (gpu_lib() does that by itself unless something loaded a runtime earlier)" % (k, rp[k]))
    return rp


def gpu_lib():
    """libkssd_gpu.so; raises if it has not been built (no fallback exists)."""
    global _gpu
    if _gpu is None:
        if not os.path.exists(GPU_LIB):
            raise ImportError("public_kssd_amd: %s is missing -- build it with `make -C public_kssd_amd` "
                              "(hipcc --offload-arch=gfx950); there is no CPU fallback" % GPU_LIB)
        if "torch" not in sys.modules and not os.environ.get("KSSD_NO_TORCH"):
            try:    # the harness uses torch next to this library: its runtime first (assert_single_runtime)
                import torch  # noqa: F401
            except ImportError:
                pass
        L = C.CDLL(GPU_LIB)
        assert_single_runtime()
        vp, u32, u64, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int
        L.kssd_gpu_strerror.restype = C.c_char_p
        L.kssd_gpu_strerror.argtypes = [i32]
        L.kssd_gpu_last_hip_error.restype = C.c_char_p
        L.kssd_gpu_create.argtypes = [C.POINTER(vp), C.POINTER(ShufHdr), vp, i32]
        L.kssd_gpu_create_compact.argtypes = [C.POINTER(vp), C.POINTER(ShufHdr), vp, u32, i32]
        L.kssd_gpu_create_for_dist.argtypes = [C.POINTER(vp), i32, i32]
        L.kssd_gpu_destroy.argtypes = [vp]
        L.kssd_gpu_destroy.restype = None
        L.kssd_gpu_get_info.argtypes = [vp, C.POINTER(GpuInfo)]
        L.kssd_gpu_sketch_device.argtypes = [vp, vp, vp, vp, u32, u32, u32, vp, vp, u64, vp]
        L.kssd_gpu_sketch_plan.argtypes = [vp, vp, vp, vp, u32, u32, u32, vp, vp, u64]
        L.kssd_gpu_sketch_phase.argtypes = [vp, i32, vp]
        L.kssd_gpu_sketch_status.argtypes = [vp, C.POINTER(u64), C.POINTER(C.c_int64), vp]
        L.kssd_gpu_sketch_batch.argtypes = [vp, vp, vp, vp, u32, u32, u32, C.POINTER(vp), C.POINTER(vp),
                                            C.POINTER(C.c_int64)]
        L.kssd_gpu_sketch_batch_pos.argtypes = [vp, vp, vp, vp, u32, u32, u32, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp),
                                                C.POINTER(C.c_int64)]
        L.kssd_gpu_sketch_set_pos_output.argtypes = [vp, vp]
        L.kssd_gpu_mask_summarise_device.argtypes = [vp, vp, u64, vp, vp]
        L.kssd_gpu_sketch_set_mask_summary.argtypes = [vp, vp]
        L.kssd_gpu_set_union.argtypes = [vp, vp, u64, C.c_int, C.POINTER(vp), C.POINTER(u64)]
        L.kssd_gpu_set_filter.argtypes = [vp, vp, vp, u32, vp, u64, C.c_int, C.POINTER(vp), C.POINTER(vp)]
        L.kssd_gpu_free.argtypes = [vp]
        L.kssd_gpu_free.restype = None
        L.kssd_gpu_index_build_device.argtypes = [vp, vp, vp, u32, u64, vp]
        L.kssd_gpu_index_status.argtypes = [vp, vp]
        L.kssd_gpu_runtime_path.restype = C.c_char_p
        L.kssd_gpu_runtime_path.argtypes = [i32]
        L.kssd_gpu_index_set_exact.argtypes = [vp, i32]
        L.kssd_gpu_dist_device.argtypes = [vp, vp, vp, u32, u32, u32, vp, vp, vp, vp, vp, vp]
        L.kssd_gpu_dist.argtypes = [vp, vp, vp, u32, vp, vp, u32, vp, vp, vp, vp, vp]
        L.kssd_gpu_dist_device_transposed.argtypes = [vp, vp, vp, u32, u32, u32, vp, u64, vp, vp, vp, vp, vp, vp]
        L.kssd_gpu_dist_counts_device.argtypes = [vp, vp, vp, u32, u32, u32, vp, vp]
        L.kssd_gpu_transpose_metrics_device.argtypes = [vp, vp, u32, u32, u32, vp, u64, vp, vp, vp, vp, vp, vp]
        L.kssd_gpu_dist_device_long.argtypes = [vp, vp, vp, u32, u32, u32, u64, vp, vp, vp, vp, vp, vp]
        L.kssd_gpu_tokenise_fasta_device.argtypes = [vp, vp, vp, vp, u32, vp, vp, vp, vp]
        L.kssd_gpu_tokenise_status.argtypes = [vp, C.POINTER(C.c_int64), vp, vp]
        L.kssd_gpu_text_reserve.argtypes = [vp, u64]
        L.kssd_gpu_text_put.argtypes = [vp, u64, vp, u64]
        L.kssd_gpu_text_put.restype = C.c_int64
        L.kssd_gpu_text_wait.argtypes = [vp, C.c_int64]
        L.kssd_gpu_tokenise_fastq_device.argtypes = [vp, vp, vp, vp, u32, vp, vp, vp, vp]
        L.kssd_gpu_tokenise_fastq_status.argtypes = [vp, C.POINTER(C.c_int64), vp, vp, vp]
        L.kssd_gpu_sketch_fastq_text.argtypes = [vp, vp, vp, vp, u32, u32, u32, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), vp,
                                                 C.POINTER(C.c_int64)]
        L.kssd_gpu_concat_units_device.argtypes = [vp, vp, vp, u32, u32, u64, vp, vp, vp]
        L.kssd_gpu_index_set_filter.argtypes = [vp, i32, u32, u32]
        L.kssd_gpu_sketch_fasta_text.argtypes = [vp, vp, vp, vp, u32, u32, u32, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp),
                                                 C.POINTER(C.c_int64)]
        L.kssd_gpu_kernel_time.argtypes = [vp, i32, i32, C.POINTER(C.c_float), C.POINTER(u32)]
        L.kssd_gpu_set_kernel_timing.argtypes = [vp, u32]
        L.kssd_gpu_kernel_times.argtypes = [vp, i32, vp, u32, C.POINTER(u32)]
        L.kssd_gpu_scan_stats.argtypes = [vp, C.POINTER(u64), C.POINTER(u64), vp]
        L.kssd_gpu_set_lds_sort_limit.argtypes = [vp, u32]
        L.kssd_gpu_set_scan_grid.argtypes = [vp, u32]
        L.kssd_gpu_dist_multi.argtypes = [vp, i32, i32, vp, vp, u32, vp, vp, u32, vp, vp, vp, vp, vp]
        L.kssd_gpu_device_count.restype = i32
        L.kssd_gpu_dist_select.argtypes = [vp, vp, vp, u32, vp, vp, u32, i32, i32, i32, C.c_double, i32, vp,
                                           C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
        L.kssd_gpu_host_alloc.restype = vp
        L.kssd_gpu_host_alloc.argtypes = [C.c_size_t]
        L.kssd_gpu_host_free.restype = None
        L.kssd_gpu_host_free.argtypes = [vp]
        _gpu = L
    return _gpu


def host_lib():
    global _host
    if _host is None:
        if not os.path.exists(HOST_LIB):
            raise ImportError("public_kssd_amd: %s is missing -- build it with `make -C public_kssd_amd`" % HOST_LIB)
        L = C.CDLL(HOST_LIB)
        vp, u32, u64, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int
        L.kssd_host_strerror.restype = C.c_char_p
        L.kssd_host_strerror.argtypes = [i32]
        L.kssd_shuf_generate.argtypes = [C.POINTER(_Shuf), i32, i32, i32, u64]
        L.kssd_shuf_write.argtypes = [C.POINTER(_Shuf), C.c_char_p]
        L.kssd_shuf_read.argtypes = [C.POINTER(_Shuf), C.c_char_p]
        L.kssd_shuf_release.argtypes = [C.POINTER(_Shuf)]
        L.kssd_shuf_release.restype = None
        L.kssd_batch_create.restype = vp
        L.kssd_batch_destroy.argtypes = [vp]
        L.kssd_batch_destroy.restype = None
        L.kssd_batch_clear.argtypes = [vp]
        L.kssd_batch_clear.restype = None
        L.kssd_batch_add_fasta.argtypes = [vp, C.c_char_p, C.c_size_t]
        L.kssd_batch_add_fastq.argtypes = [vp, C.c_char_p, C.c_size_t, i32, C.POINTER(u64)]
        L.kssd_batch_add_reads.argtypes = [vp, C.c_char_p, C.c_size_t, C.POINTER(u64)]
        L.kssd_batch_add_file.argtypes = [vp, C.c_char_p, i32, i32, C.POINTER(u64)]
        L.kssd_batch_add_fasta_reads.argtypes = [vp, C.c_char_p, C.c_size_t, C.POINTER(vp), C.POINTER(u64)]
        L.kssd_byread_write.argtypes = [C.c_char_p, u32, i32, i32, C.c_char_p, u32, vp, vp, u64, vp, u64]
        L.kssd_host_free.argtypes = [vp]
        L.kssd_host_free.restype = None
        for f in ("packed", "mask", "chunk_off"):
            getattr(L, "kssd_batch_" + f).restype = vp
            getattr(L, "kssd_batch_" + f).argtypes = [vp]
        L.kssd_batch_n_chunks.restype = u64
        L.kssd_batch_n_chunks.argtypes = [vp]
        L.kssd_batch_n_genomes.restype = u32
        L.kssd_batch_n_genomes.argtypes = [vp]
        L.kssd_batch_n_positions.restype = u64
        L.kssd_batch_n_positions.argtypes = [vp, u32]
        L.kssd_batch_append.argtypes = [vp, vp]
        L.kssd_batch_create_ex.restype = vp
        L.kssd_batch_create_ex.argtypes = [vp, vp]
        L.kssd_batch_reserve.argtypes = [vp, u32, vp, C.POINTER(u32)]
        L.kssd_batch_fill_text.argtypes = [vp, u32, i32, C.c_char_p, C.c_size_t, i32, C.POINTER(u64)]
        L.kssd_derive.argtypes = [C.POINTER(_Derived), i32, i32, i32]
        L.kssd_sketchset_release.argtypes = [C.POINTER(_SketchSet)]
        L.kssd_sketchset_release.restype = None
        L.kssd_slot_order_pos64.argtypes = [vp, vp, u64, u32]
        L.kssd_slot_order_pos64.restype = C.c_int
        L.kssd_slot_order.argtypes = [vp, u64, u32]
        L.kssd_slot_order.restype = C.c_int
        L.kssd_slot_order_pos.argtypes = [vp, vp, u64, u32]
        L.kssd_slot_order_pos.restype = C.c_int
        L.kssd_sketchset_write.argtypes = [C.POINTER(_SketchSet), C.c_char_p, u32, i32]
        L.kssd_sketchset_read.argtypes = [C.POINTER(_SketchSet), C.c_char_p]
        L.kssd_index_write.argtypes = [C.POINTER(_SketchSet), C.c_char_p]
        L.kssd_index_read.argtypes = [C.POINTER(_SketchSet), C.c_char_p]
        L.kssd_probe_dir.argtypes = [C.c_char_p]
        L.kssd_distance_print.argtypes = [C.c_char_p, vp, C.POINTER(_SketchSet), C.POINTER(_SketchSet),
                                          C.POINTER(_PrintOpt)]
        L.kssd_distance_print_pairs.argtypes = [C.c_char_p, vp, vp, vp, C.POINTER(_SketchSet), C.POINTER(_SketchSet),
                                                C.POINTER(_PrintOpt)]
        _host = L
    return _host


def _hck(rc):
    if rc != 0:
        raise KssdError(rc, host_lib().kssd_host_strerror(rc).decode())


def _gck(rc):
    if rc != 0:
        raise KssdError(rc, gpu_lib().kssd_gpu_strerror(rc).decode())


# ------------------------------------------------------------------------------------------------------
# .shuf
# ------------------------------------------------------------------------------------------------------
class Shuf:
    """A dimension-reduction shuffle (the reference's dim_shuffle_t): header + int32 permutation."""

    def __init__(self, hdr, table):
        self.id, self.k, self.subk, self.drlevel = hdr
        self.table = np.ascontiguousarray(table, dtype=np.int32)

    @classmethod
    def generate(cls, k, subk, drlevel, seed):
        s = _Shuf()
        _hck(host_lib().kssd_shuf_generate(C.byref(s), k, subk, drlevel, seed))
        try:
            t = np.ctypeslib.as_array(s.table, shape=(16 ** subk,)).copy()
            return cls((s.id, s.k, s.subk, s.drlevel), t)
        finally:
            host_lib().kssd_shuf_release(C.byref(s))

    @classmethod
    def read(cls, path):
        s = _Shuf()
        _hck(host_lib().kssd_shuf_read(C.byref(s), os.fsencode(path)))
        try:
            t = np.ctypeslib.as_array(s.table, shape=(16 ** s.subk,)).copy()
            return cls((s.id, s.k, s.subk, s.drlevel), t)
        finally:
            host_lib().kssd_shuf_release(C.byref(s))

    @staticmethod
    def read_core(path):
        """(header tuple, accepted u32[dim_end], from_cache) -- what the command line loads of a .shuf (kssd_shuf_read_core)"""
        s = _Shuf()
        acc, n, cached = C.c_void_p(), C.c_uint32(0), C.c_int(0)
        _hck(host_lib().kssd_shuf_read_core(os.fsencode(path), C.byref(s), C.byref(acc), C.byref(n), C.byref(cached)))
        try:
            a = np.frombuffer((C.c_char * (4 * n.value)).from_address(acc.value), dtype=np.uint32).copy()
        finally:
            C.CDLL(None).free(acc)
        return (s.id, s.k, s.subk, s.drlevel), a, bool(cached.value)

    def write(self, path):
        s = _Shuf(self.id, self.k, self.subk, self.drlevel, self.table.ctypes.data_as(C.POINTER(C.c_int32)))
        _hck(host_lib().kssd_shuf_write(C.byref(s), os.fsencode(path)))

    def hdr(self):
        return ShufHdr(self.id, self.k, self.subk, self.drlevel)


# ------------------------------------------------------------------------------------------------------
# on-disk sketch / index formats and the distance report (host C, kssd_formats.c)
# ------------------------------------------------------------------------------------------------------
def slot_order_pos64(tuples, first_pos, hashsize):
    """tuples of more than 32 bits (k - drlevel = 9) in the reference's file order (kssd_slot_order_pos64)"""
    t = np.ascontiguousarray(tuples, dtype=np.uint64).copy()
    p = np.ascontiguousarray(first_pos, dtype=np.uint32)
    _hck(host_lib().kssd_slot_order_pos64(t.ctypes.data, p.ctypes.data, len(t), hashsize))
    return t


def derive(k, subk, drlevel):
    d = _Derived()
    _hck(host_lib().kssd_derive(C.byref(d), k, subk, drlevel))
    return d


class SketchSet:
    """Sketches of n genomes: CSR of full reduced tuples + the header fields of cofiles.stat."""

    def __init__(self, shuf_id, kmerlen, dim_rd_len, comp_num, names, off, ids, counts=None, sub=None):
        self.shuf_id, self.kmerlen, self.dim_rd_len, self.comp_num = shuf_id, kmerlen, dim_rd_len, comp_num
        # k - drlevel = 9 (256 components): the tuples' low four bits, ids = tuple >> 4 (write only)
        self.sub = None if sub is None else np.ascontiguousarray(sub, dtype=np.uint8).copy()
        self.names = list(names)
        self.off = np.ascontiguousarray(off, dtype=np.uint64)
        self.ids = np.ascontiguousarray(ids, dtype=np.uint32).copy()
        # abundance sketches (dist -A): occurrences of ids[i], u16; None for plain sketches
        self.counts = None if counts is None else np.ascontiguousarray(counts, dtype=np.uint16).copy()
        assert len(self.off) == len(self.names) + 1
        assert self.counts is None or len(self.counts) == len(self.ids)

    def _c(self):
        nm = b"".join(os.fsencode(n).ljust(256, b"\0")[:256] for n in self.names) or b"\0"
        if getattr(self, "_nm", None) != nm:  # (a set passed as both arguments of one call must not replace the buffer the first struct points into)
            self._nm = nm
            self._nmbuf = C.create_string_buffer(nm, len(nm))
        self._ids = self.ids if len(self.ids) else np.zeros(1, np.uint32)
        koc = self.counts is not None
        self._cnt = (self.counts if len(self.counts) else np.zeros(1, np.uint16)) if koc else None
        return _SketchSet(self.shuf_id, int(koc), self.kmerlen, self.dim_rd_len, self.comp_num, len(self.names),
                          self.off.ctypes.data, self._ids.ctypes.data, C.addressof(self._nmbuf),
                          self._cnt.ctypes.data if koc else None,
                          (self.sub if len(self.sub) else np.zeros(1, np.uint8)).ctypes.data if self.sub is not None else None)

    @classmethod
    def _from_c(cls, s):
        n = s.n
        off = np.frombuffer((C.c_char * (8 * (n + 1))).from_address(s.off), dtype=np.uint64).copy()
        tot = int(off[-1])
        ids = (np.frombuffer((C.c_char * (4 * tot)).from_address(s.ids), dtype=np.uint32).copy() if tot
               else np.zeros(0, np.uint32))
        raw = (C.c_char * (256 * n)).from_address(s.names).raw if n else b""
        names = [raw[256 * i:256 * (i + 1)].split(b"\0")[0].decode() for i in range(n)]
        counts = None
        if s.koc and s.counts:
            counts = (np.frombuffer((C.c_char * (2 * tot)).from_address(s.counts), dtype=np.uint16).copy() if tot
                      else np.zeros(0, np.uint16))
        return cls(s.shuf_id, s.kmerlen, s.dim_rd_len, s.comp_num, names, off, ids, counts)

    def write(self, d, hashsize, slot_order=True):
        """cofiles.stat + combco.<c> + combco.index.<c>"""
        _hck(host_lib().kssd_sketchset_write(C.byref(self._c()), os.fsencode(d), hashsize, int(slot_order)))

    def write_index(self, d):
        """mcofiles.stat + mco.index.<c> (dense, 2 GiB per component) + mco.<c>"""
        _hck(host_lib().kssd_index_write(C.byref(self._c()), os.fsencode(d)))

    @classmethod
    def read(cls, d):
        s = _SketchSet()
        _hck(host_lib().kssd_sketchset_read(C.byref(s), os.fsencode(d)))
        try:
            return cls._from_c(s)
        finally:
            host_lib().kssd_sketchset_release(C.byref(s))

    @classmethod
    def read_index(cls, d):
        s = _SketchSet()
        _hck(host_lib().kssd_index_read(C.byref(s), os.fsencode(d)))
        try:
            return cls._from_c(s)
        finally:
            host_lib().kssd_sketchset_release(C.byref(s))

    def sets_by_name(self):
        return {os.path.basename(n): np.sort(self.ids[int(self.off[i]):int(self.off[i + 1])])
                for i, n in enumerate(self.names)}


def byread_write(d, shuf, fname, ids, pos, read_start):
    """combco.<c> / combco.index.<c> / cofiles.stat of `kssd dist --byread` for one file (reads2mco)"""
    ids = np.ascontiguousarray(ids, dtype=np.uint32)
    pos = np.ascontiguousarray(pos, dtype=np.uint32)
    rs = np.ascontiguousarray(read_start, dtype=np.uint64)
    name = os.fsencode(fname)[:255].ljust(256, b"\0")
    _hck(host_lib().kssd_byread_write(os.fsencode(d), shuf.id & 0xFFFFFFFF, shuf.k, shuf.drlevel, name, 1,
                                      ids.ctypes.data, pos.ctypes.data, len(ids), rs.ctypes.data, len(rs)))


def slot_order_pos(ids, first_pos, hashsize):
    """ids of one genome in the reference's file order, insertions replayed in sequence order"""
    a = np.ascontiguousarray(ids, dtype=np.uint32).copy()
    p = np.ascontiguousarray(first_pos, dtype=np.uint32)
    _hck(host_lib().kssd_slot_order_pos(a.ctypes.data, p.ctypes.data, len(a), hashsize))
    return a


def slot_order(ids, hashsize):
    a = np.ascontiguousarray(ids, dtype=np.uint32).copy()
    _hck(host_lib().kssd_slot_order(a.ctypes.data, len(a), hashsize))
    return a


def distance_print(path, shared, ref, qry, metric=0, pfield=2, correction=0, dthreshold=1.0, n_max=0, threads=1):
    shared = np.ascontiguousarray(shared, dtype=np.uint32)
    assert shared.shape == (len(qry.names), len(ref.names))
    o = _PrintOpt(metric, pfield, correction, dthreshold, n_max, threads)
    _hck(host_lib().kssd_distance_print(os.fsencode(path), shared.ctypes.data, C.byref(ref._c()), C.byref(qry._c()),
                                        C.byref(o)))


def distance_print_pairs(path, pair_off, pair_ref, pair_shared, ref, qry, metric=0, pfield=2, correction=0, dthreshold=1.0,
                         n_max=0, threads=1):
    """the report from the candidate pairs of GpuCtx.dist_select"""
    po = np.ascontiguousarray(pair_off, dtype=np.uint64)
    pr = np.ascontiguousarray(pair_ref, dtype=np.uint32)
    ps = np.ascontiguousarray(pair_shared, dtype=np.uint32)
    if len(pr) == 0:
        pr, ps = np.zeros(1, np.uint32), np.zeros(1, np.uint32)
    o = _PrintOpt(metric, pfield, correction, dthreshold, n_max, threads)
    _hck(host_lib().kssd_distance_print_pairs(os.fsencode(path), po.ctypes.data, pr.ctypes.data, ps.ctypes.data,
                                              C.byref(ref._c()), C.byref(qry._c()), C.byref(o)))


# ------------------------------------------------------------------------------------------------------
# packed batches
# ------------------------------------------------------------------------------------------------------
class Batch:
    """Genomes tokenised into the packed device layout (2-bit bases + validity mask, 4096-base chunks)."""

    def __init__(self, pinned=False):
        """pinned: packed / mask arrays in page-locked memory of the GPU runtime (kssd_gpu_host_alloc)"""
        if pinned:
            g = gpu_lib()
            self.h = host_lib().kssd_batch_create_ex(C.cast(g.kssd_gpu_host_alloc, C.c_void_p),
                                                     C.cast(g.kssd_gpu_host_free, C.c_void_p))
        else:
            self.h = host_lib().kssd_batch_create()
        if not self.h:
            raise MemoryError

    def close(self):
        if self.h:
            host_lib().kssd_batch_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def clear(self):
        host_lib().kssd_batch_clear(self.h)

    def add_fasta(self, text):
        text = bytes(text)
        _hck(host_lib().kssd_batch_add_fasta(self.h, text, len(text)))

    def add_fastq(self, text, Q=0):
        text = bytes(text)
        n = C.c_uint64(0)
        _hck(host_lib().kssd_batch_add_fastq(self.h, text, len(text), Q, C.byref(n)))
        return n.value

    def add_reads(self, text):
        """reads framed like dist -A frames them; returns the number of reads"""
        text = bytes(text)
        n = C.c_uint64(0)
        _hck(host_lib().kssd_batch_add_reads(self.h, text, len(text), C.byref(n)))
        return n.value

    def add_fasta_reads(self, text):
        """one FASTA file as ONE genome for dist --byread; returns the cut points of its reads (uint64[number of '>'])"""
        text = bytes(text)
        p, n = C.c_void_p(), C.c_uint64(0)
        _hck(host_lib().kssd_batch_add_fasta_reads(self.h, text, len(text), C.byref(p), C.byref(n)))
        try:
            return (np.frombuffer((C.c_char * (8 * n.value)).from_address(p.value), dtype=np.uint64).copy()
                    if n.value else np.zeros(0, np.uint64))
        finally:
            host_lib().kssd_host_free(p)

    def reserve(self, max_positions):
        """append len(max_positions) empty genomes with that much room each; returns the index of the first"""
        mp = np.ascontiguousarray(max_positions, dtype=np.uint64)
        first = C.c_uint32(0)
        _hck(host_lib().kssd_batch_reserve(self.h, len(mp), mp.ctypes.data, C.byref(first)))
        return first.value

    def fill_text(self, genome, text, kind=0, Q=0):
        """tokenise text into a reserved genome (thread-safe across genomes); kind 0 FASTA, 1 FASTQ, 2 reads of -A"""
        text = bytes(text)
        n = C.c_uint64(0)
        _hck(host_lib().kssd_batch_fill_text(self.h, genome, kind, text, len(text), Q, C.byref(n)))
        return n.value

    def add_file(self, path, is_fastq=False, Q=0):
        n = C.c_uint64(0)
        _hck(host_lib().kssd_batch_add_file(self.h, os.fsencode(path), int(is_fastq), Q, C.byref(n)))
        return n.value

    @property
    def n_chunks(self):
        return host_lib().kssd_batch_n_chunks(self.h)

    @property
    def n_genomes(self):
        return host_lib().kssd_batch_n_genomes(self.h)

    def n_positions(self, g):
        return host_lib().kssd_batch_n_positions(self.h, g)

    def _view(self, ptr, n, dtype):
        if n == 0 or not ptr:
            return np.zeros(0, dtype=dtype)
        buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
        return np.frombuffer(buf, dtype=dtype, count=n)

    def packed(self):
        """view incl. the slack words (valid until the batch is modified)"""
        return self._view(host_lib().kssd_batch_packed(self.h), self.n_chunks * CHUNK_WORDS + SLACK_WORDS, np.uint32)

    def mask(self):
        return self._view(host_lib().kssd_batch_mask(self.h), self.n_chunks * CHUNK_MASKW + SLACK_WORDS, np.uint32)

    def chunk_off(self):
        return self._view(host_lib().kssd_batch_chunk_off(self.h), self.n_genomes + 1, np.uint64).copy()


# ------------------------------------------------------------------------------------------------------
# the device context
# ------------------------------------------------------------------------------------------------------
def device_count():
    return gpu_lib().kssd_gpu_device_count()


def dist_multi(devices, kmerlen, roff, rids, qoff, qids, planes=True):
    """kssd_gpu_dist_multi: the query rows in len(devices) contiguous blocks, one per device (entries may repeat)"""
    roff = np.ascontiguousarray(roff, dtype=np.uint64)
    qoff = np.ascontiguousarray(qoff, dtype=np.uint64)
    rids = np.ascontiguousarray(rids, dtype=np.uint32)
    qids = np.ascontiguousarray(qids, dtype=np.uint32)
    R, Q = len(roff) - 1, len(qoff) - 1
    devs = (C.c_int * len(devices))(*devices)
    shared = np.zeros((Q, R), dtype=np.uint32)
    pl = [np.zeros((Q, R), dtype=np.float64) for _ in range(4)] if planes else [None] * 4
    _gck(gpu_lib().kssd_gpu_dist_multi(devs, len(devices), kmerlen, roff.ctypes.data, rids.ctypes.data, R, qoff.ctypes.data,
                                       qids.ctypes.data, Q, shared.ctypes.data, *[_ptr(p) for p in pl]))
    return (shared, *pl) if planes else shared


def _ptr(x):
    """device or host address of a numpy array / torch tensor / int / None"""
    if x is None:
        return None
    if isinstance(x, int):
        return x
    if isinstance(x, np.ndarray):
        return x.ctypes.data
    return x.data_ptr()  # torch tensor


class GpuCtx:
    """kssd_gpu_ctx: device tables for one .shuf + workspaces (one calling thread per context)."""

    def __init__(self, shuf=None, device=0, kmerlen=None):
        """shuf: a Shuf (sketch + distance context); or kmerlen=2k for a distance-only context"""
        self.h = C.c_void_p()
        if shuf is None:
            _gck(gpu_lib().kssd_gpu_create_for_dist(C.byref(self.h), kmerlen, device))
        else:
            hdr = shuf.hdr()
            _gck(gpu_lib().kssd_gpu_create(C.byref(self.h), C.byref(hdr), shuf.table.ctypes.data, device))
        self.info = GpuInfo()
        _gck(gpu_lib().kssd_gpu_get_info(self.h, C.byref(self.info)))
        if DEFAULT_LDS_SORT_LIMIT:
            self.set_lds_sort_limit(DEFAULT_LDS_SORT_LIMIT)

    def close(self):
        if getattr(self, "h", None):
            gpu_lib().kssd_gpu_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            if not sys.is_finalizing():  # never call into HIP while the interpreter (and the runtime) shut down
                self.close()
        except Exception:
            pass

    # host-level ---------------------------------------------------------------------------------------
    def sketch_batch(self, batch, flags=SKETCH_FASTA, min_occ=1):
        """(off uint64[n+1], ids uint32) -- ascending distinct ids per genome"""
        return self.sketch_packed(batch.packed(), batch.mask(), batch.chunk_off(), flags, min_occ)

    def sketch_batch_pos(self, batch, flags=SKETCH_FASTA, min_occ=1):
        """(off, ids, first_pos): as sketch_batch plus every id's first position inside its genome"""
        chunk_off = np.ascontiguousarray(batch.chunk_off(), dtype=np.uint64)
        n = len(chunk_off) - 1
        po, pi, pp, bad = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_int64(-1)
        rc = gpu_lib().kssd_gpu_sketch_batch_pos(self.h, _ptr(batch.packed()), _ptr(batch.mask()), chunk_off.ctypes.data, n,
                                                 flags, min_occ, C.byref(po), C.byref(pi), C.byref(pp), C.byref(bad))
        if rc != 0:
            e = KssdError(rc, gpu_lib().kssd_gpu_strerror(rc).decode())
            e.bad_genome = bad.value
            raise e
        try:
            off = np.frombuffer((C.c_char * (8 * (n + 1))).from_address(po.value), dtype=np.uint64).copy()
            tot = int(off[-1])
            ids = (np.frombuffer((C.c_char * (4 * tot)).from_address(pi.value), dtype=np.uint32).copy()
                   if tot else np.zeros(0, np.uint32))
            pos = (np.frombuffer((C.c_char * (4 * tot)).from_address(pp.value), dtype=np.uint32).copy()
                   if tot else np.zeros(0, np.uint32))
        finally:
            for q in (po, pi, pp):
                gpu_lib().kssd_gpu_free(q)
        return off, ids, pos

    def sketch_packed(self, packed, mask, chunk_off, flags=SKETCH_FASTA, min_occ=1):
        chunk_off = np.ascontiguousarray(chunk_off, dtype=np.uint64)
        n = len(chunk_off) - 1
        po, pi, bad = C.c_void_p(), C.c_void_p(), C.c_int64(-1)
        rc = gpu_lib().kssd_gpu_sketch_batch(self.h, _ptr(packed), _ptr(mask), chunk_off.ctypes.data, n, flags,
                                             min_occ, C.byref(po), C.byref(pi), C.byref(bad))
        if rc != 0:
            e = KssdError(rc, gpu_lib().kssd_gpu_strerror(rc).decode())
            e.bad_genome = bad.value
            raise e
        try:
            off = np.frombuffer((C.c_char * (8 * (n + 1))).from_address(po.value), dtype=np.uint64).copy()
            tot = int(off[-1])
            ids = (np.frombuffer((C.c_char * (4 * tot)).from_address(pi.value), dtype=np.uint32).copy()
                   if tot else np.zeros(0, np.uint32))
        finally:
            gpu_lib().kssd_gpu_free(po)
            gpu_lib().kssd_gpu_free(pi)
        return off, ids

    @staticmethod
    def _text_layout(texts):
        """the files one after the other, each on a 16-byte boundary: (buffer uint8, off uint64, len uint64)"""
        lens = np.array([len(t) for t in texts], dtype=np.uint64)
        offs = np.zeros(len(texts), dtype=np.uint64)
        at = 0
        for i, t in enumerate(texts):
            offs[i] = at
            at += (len(t) + 15) // 16 * 16
        buf = np.zeros(max(at, 16), dtype=np.uint8)
        for i, t in enumerate(texts):
            buf[int(offs[i]):int(offs[i]) + len(t)] = np.frombuffer(bytes(t), dtype=np.uint8)
        return buf, offs, lens

    def sketch_fastq_texts(self, texts, flags=SKETCH_FASTA, min_occ=1, with_pos=False):
        """FASTQ texts (-Q 0) tokenised ON THE DEVICE and sketched: (off, ids[, pos], lines per text).  KssdError with
        code KSSD_ERR_UNSUPPORTED and .bad_genome when a text needs the host tokeniser."""
        return self.sketch_fasta_texts(texts, flags, min_occ, with_pos, _fastq=True)

    def sketch_fasta_texts(self, texts, flags=SKETCH_FASTA, min_occ=1, with_pos=False, _fastq=False):
        """FASTA texts tokenised ON THE DEVICE and sketched: (off, ids[, pos]); one genome per text"""
        buf, offs, lens = self._text_layout(texts)
        n = len(texts)
        self._last_n = n
        po, pi, pp, bad = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_int64(-1)
        lines = np.zeros(max(n, 1), dtype=np.uint64)
        if _fastq:
            rc = gpu_lib().kssd_gpu_sketch_fastq_text(self.h, buf.ctypes.data, offs.ctypes.data, lens.ctypes.data, n, flags, min_occ,
                                                      C.byref(po), C.byref(pi), C.byref(pp) if with_pos else None, lines.ctypes.data,
                                                      C.byref(bad))
        else:
            rc = gpu_lib().kssd_gpu_sketch_fasta_text(self.h, buf.ctypes.data, offs.ctypes.data, lens.ctypes.data, n, flags, min_occ,
                                                      C.byref(po), C.byref(pi), C.byref(pp) if with_pos else None, C.byref(bad))
        if rc != 0:
            e = KssdError(rc, gpu_lib().kssd_gpu_strerror(rc).decode())
            e.bad_genome = bad.value
            raise e
        try:
            off = np.frombuffer((C.c_char * (8 * (n + 1))).from_address(po.value), dtype=np.uint64).copy()
            tot = int(off[-1])
            ids = (np.frombuffer((C.c_char * (4 * tot)).from_address(pi.value), dtype=np.uint32).copy() if tot else np.zeros(0, np.uint32))
            pos = None
            if with_pos:
                pos = (np.frombuffer((C.c_char * (4 * tot)).from_address(pp.value), dtype=np.uint32).copy() if tot else np.zeros(0, np.uint32))
        finally:
            for q in (po, pi, pp):
                if q.value:
                    gpu_lib().kssd_gpu_free(q)
        res = (off, ids, pos) if with_pos else (off, ids)
        return res + (lines[:n],) if _fastq else res

    def tokenise_fasta_device(self, d_text, text_off, text_len, d_packed, d_mask, chunk_off, stream=None, fastq=False, status=True, d_summary=None):
        """device-level: raw FASTA (or FASTQ) bytes in HBM -> packed batch in HBM; returns (rc, bad_file, positions per
        file[, lines per file]); status=False: the kernels are enqueued and nothing is synchronised (returns None)"""
        to = np.ascontiguousarray(text_off, dtype=np.uint64)
        tl = np.ascontiguousarray(text_len, dtype=np.uint64)
        co = np.ascontiguousarray(chunk_off, dtype=np.uint64)
        if d_summary is not None and not fastq:   # (the mask's summary words written beside the mask: u64 per chunk, device)
            L = gpu_lib()
            L.kssd_gpu_tokenise_fasta_device_summary.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p,
                                                                 C.c_void_p, C.c_void_p, C.c_void_p]
            _gck(L.kssd_gpu_tokenise_fasta_device_summary(self.h, _ptr(d_text), to.ctypes.data, tl.ctypes.data, len(tl), _ptr(d_packed), _ptr(d_mask),
                                                          _ptr(d_summary), co.ctypes.data, stream))
        else:
            fn = gpu_lib().kssd_gpu_tokenise_fastq_device if fastq else gpu_lib().kssd_gpu_tokenise_fasta_device
            _gck(fn(self.h, _ptr(d_text), to.ctypes.data, tl.ctypes.data, len(tl), _ptr(d_packed), _ptr(d_mask), co.ctypes.data, stream))
        if not status:
            return None
        bad = C.c_int64(-1)
        npos = np.zeros(max(len(tl), 1), dtype=np.uint64)
        if fastq:
            nlines = np.zeros(max(len(tl), 1), dtype=np.uint64)
            rc = gpu_lib().kssd_gpu_tokenise_fastq_status(self.h, C.byref(bad), npos.ctypes.data, nlines.ctypes.data, stream)
            return rc, bad.value, npos[:len(tl)], nlines[:len(tl)]
        rc = gpu_lib().kssd_gpu_tokenise_status(self.h, C.byref(bad), npos.ctypes.data, stream)
        return rc, bad.value, npos[:len(tl)]

    def tokenise_fastq_device(self, d_text, text_off, text_len, d_packed, d_mask, chunk_off, stream=None):
        return self.tokenise_fasta_device(d_text, text_off, text_len, d_packed, d_mask, chunk_off, stream, fastq=True)

    def dist(self, roff, rids, qoff, qids, planes=True):
        """shared uint32[Q,R] (+ J, MashD, C, AafD float64[Q,R] when planes)"""
        roff = np.ascontiguousarray(roff, dtype=np.uint64)
        qoff = np.ascontiguousarray(qoff, dtype=np.uint64)
        rids = np.ascontiguousarray(rids, dtype=np.uint32)
        qids = np.ascontiguousarray(qids, dtype=np.uint32)
        R, Q = len(roff) - 1, len(qoff) - 1
        shared = np.zeros((Q, R), dtype=np.uint32)
        pl = [np.zeros((Q, R), dtype=np.float64) for _ in range(4)] if planes else [None] * 4
        _gck(gpu_lib().kssd_gpu_dist(self.h, roff.ctypes.data, rids.ctypes.data, R, qoff.ctypes.data,
                                     qids.ctypes.data, Q, shared.ctypes.data, *[_ptr(p) for p in pl]))
        return (shared, *pl) if planes else shared

    def dist_select(self, roff, rids, qoff, qids, metric=0, correction=0, dim_rd_len=0, dthreshold=1.0, n_max=0, dense=False):
        """(pair_off uint64[Q+1], pair_ref, pair_shared[, shared uint32[Q,R]]): the pairs that can appear in the report"""
        roff = np.ascontiguousarray(roff, dtype=np.uint64)
        qoff = np.ascontiguousarray(qoff, dtype=np.uint64)
        rids = np.ascontiguousarray(rids, dtype=np.uint32)
        qids = np.ascontiguousarray(qids, dtype=np.uint32)
        R, Q = len(roff) - 1, len(qoff) - 1
        shared = np.zeros((Q, R), dtype=np.uint32) if dense else None
        po, pr, ps = C.c_void_p(), C.c_void_p(), C.c_void_p()
        _gck(gpu_lib().kssd_gpu_dist_select(self.h, roff.ctypes.data, rids.ctypes.data, R, qoff.ctypes.data, qids.ctypes.data, Q,
                                            metric, correction, dim_rd_len, dthreshold, n_max, _ptr(shared),
                                            C.byref(po), C.byref(pr), C.byref(ps)))
        try:
            off = np.frombuffer((C.c_char * (8 * (Q + 1))).from_address(po.value), dtype=np.uint64).copy()
            n = int(off[-1])
            ref = (np.frombuffer((C.c_char * (4 * n)).from_address(pr.value), dtype=np.uint32).copy() if n else np.zeros(0, np.uint32))
            sh = (np.frombuffer((C.c_char * (4 * n)).from_address(ps.value), dtype=np.uint32).copy() if n else np.zeros(0, np.uint32))
        finally:
            for q in (po, pr, ps):
                gpu_lib().kssd_gpu_free(q)
        return (off, ref, sh, shared) if dense else (off, ref, sh)

    # device-level (torch tensors or raw addresses; nothing is synchronised) ------------------------------
    def sketch_device(self, d_packed, d_mask, chunk_off, d_out_off, d_out_ids, out_cap, flags=SKETCH_FASTA,
                      min_occ=1, stream=None, d_summary=None):
        chunk_off = np.ascontiguousarray(chunk_off, dtype=np.uint64)
        if d_summary is not None:
            _gck(gpu_lib().kssd_gpu_sketch_set_mask_summary(self.h, _ptr(d_summary)))
        _gck(gpu_lib().kssd_gpu_sketch_device(self.h, _ptr(d_packed), _ptr(d_mask), chunk_off.ctypes.data,
                                              len(chunk_off) - 1, flags, min_occ, _ptr(d_out_off), _ptr(d_out_ids),
                                              out_cap, stream))

    def mask_summarise_device(self, d_mask, n_chunks, d_summary, stream=None):
        """d_summary[n_chunks] (u64, device) from d_mask: bit l of word c = the 64 positions of lane l of chunk c are all bases"""
        _gck(gpu_lib().kssd_gpu_mask_summarise_device(self.h, _ptr(d_mask), n_chunks, _ptr(d_summary), stream))

    def sketch_plan(self, d_packed, d_mask, chunk_off, d_out_off, d_out_ids, out_cap, flags=SKETCH_FASTA, min_occ=1, d_summary=None):
        """host part of sketch_device; the phases follow with sketch_phase (PHASE_PREP .. PHASE_FINISH, one stream).
        d_summary: the mask's summary words (mask_summarise_device) -- the scan then does not stream the mask"""
        chunk_off = np.ascontiguousarray(chunk_off, dtype=np.uint64)
        if d_summary is not None:
            _gck(gpu_lib().kssd_gpu_sketch_set_mask_summary(self.h, _ptr(d_summary)))
        _gck(gpu_lib().kssd_gpu_sketch_plan(self.h, _ptr(d_packed), _ptr(d_mask), chunk_off.ctypes.data,
                                            len(chunk_off) - 1, flags, min_occ, _ptr(d_out_off), _ptr(d_out_ids), out_cap))

    def sketch_phase(self, phase, stream=None):
        _gck(gpu_lib().kssd_gpu_sketch_phase(self.h, phase, stream))

    def sketch_status(self, stream=None):
        """(rc, total_ids, bad_genome) after synchronising the stream"""
        tot, bad = C.c_uint64(0), C.c_int64(-1)
        rc = gpu_lib().kssd_gpu_sketch_status(self.h, C.byref(tot), C.byref(bad), stream)
        return rc, tot.value, bad.value

    def index_build_device(self, d_roff, d_rids, n_ref, max_ref_ids, stream=None, check=True):
        """check (the default): the build's status is read back (this synchronises the stream) and a capped build that overflowed --
        ids that do not spread over the buckets: a database of near-identical genomes -- is repeated with exact counts, so that the
        index in place is whole when the call returns.  check=False: nothing is synchronised (a timed loop); the caller then polls
        index_status() itself -- the rows of an index that is not whole come back as 0xFFFFFFFF counts, never as stale data."""
        _gck(gpu_lib().kssd_gpu_index_build_device(self.h, _ptr(d_roff), _ptr(d_rids), n_ref, max_ref_ids, stream))
        if check:
            st = self.index_status(stream)
            if st == ERR_OVERFLOW:
                _gck(gpu_lib().kssd_gpu_index_build_device(self.h, _ptr(d_roff), _ptr(d_rids), n_ref, max_ref_ids, stream))
                st = self.index_status(stream)
            _gck(st)  # (ERR_PARAM: max_ref_ids -- the TOTAL of the references' ids -- was no upper bound of d_roff[n_ref])

    def index_status(self, stream=None):
        """synchronises the stream; 0, ERR_OVERFLOW when the (capped) build in place met a bucket fuller than its run: build again --
        or ERR_PARAM when the build's max_ref_ids was smaller than d_roff[n_ref] (the index then holds the first max_ref_ids entries)"""
        return gpu_lib().kssd_gpu_index_status(self.h, stream)

    def index_set_exact(self, exact=True):
        _gck(gpu_lib().kssd_gpu_index_set_exact(self.h, 1 if exact else 0))

    def dist_device(self, d_qoff, d_qids, n_qry, q_begin, q_end, d_shared, d_j=None, d_m=None, d_c=None, d_a=None,
                    stream=None, max_row_ids=0):
        """max_row_ids: an upper bound of the longest query row's ids (0 = rows of ordinary sketch size)"""
        if max_row_ids:
            _gck(gpu_lib().kssd_gpu_dist_device_long(self.h, _ptr(d_qoff), _ptr(d_qids), n_qry, q_begin, q_end, max_row_ids,
                                                     _ptr(d_shared), _ptr(d_j), _ptr(d_m), _ptr(d_c), _ptr(d_a), stream))
            return
        _gck(gpu_lib().kssd_gpu_dist_device(self.h, _ptr(d_qoff), _ptr(d_qids), n_qry, q_begin, q_end, _ptr(d_shared),
                                            _ptr(d_j), _ptr(d_m), _ptr(d_c), _ptr(d_a), stream))

    def dist_device_transposed(self, d_qoff, d_qids, n_qry, q_begin, q_end, d_work, out_pitch, d_shared_t, d_j=None, d_m=None, d_c=None,
                               d_a=None, stream=None):
        """rows [q_begin, q_end) against the index, the block written transposed: element (indexed sketch r, query q) at
        r * out_pitch + (q - q_begin) of every output; d_work: u32[(q_end - q_begin) x n_ref] scratch"""
        _gck(gpu_lib().kssd_gpu_dist_device_transposed(self.h, _ptr(d_qoff), _ptr(d_qids), n_qry, q_begin, q_end, _ptr(d_work), out_pitch,
                                                       _ptr(d_shared_t), _ptr(d_j), _ptr(d_m), _ptr(d_c), _ptr(d_a), stream))

    def dist_counts_device(self, d_qoff, d_qids, n_qry, q_begin, q_end, d_counts, stream=None):
        """shared counts of rows [q_begin, q_end), row-major by query (rows behind the negative filter are walked flat)"""
        _gck(gpu_lib().kssd_gpu_dist_counts_device(self.h, _ptr(d_qoff), _ptr(d_qids), n_qry, q_begin, q_end, _ptr(d_counts), stream))

    def transpose_metrics_device(self, d_qoff, n_qry, q_begin, q_end, d_counts, out_pitch, d_shared_t, d_j=None, d_m=None, d_c=None, d_a=None,
                                 stream=None):
        """counts of rows [q_begin, q_end) -> the five outputs, element (indexed sketch r, query q) at r * out_pitch + (q - q_begin)"""
        _gck(gpu_lib().kssd_gpu_transpose_metrics_device(self.h, _ptr(d_qoff), n_qry, q_begin, q_end, _ptr(d_counts), out_pitch, _ptr(d_shared_t),
                                                         _ptr(d_j), _ptr(d_m), _ptr(d_c), _ptr(d_a), stream))

    def set_union(self, ids, uniq=False):
        """ascending distinct ids (uniq: the ids that occur exactly once) -- kssd set -u / -q"""
        a = np.ascontiguousarray(ids, dtype=np.uint32)
        po, n = C.c_void_p(), C.c_uint64(0)
        _gck(gpu_lib().kssd_gpu_set_union(self.h, a.ctypes.data, len(a), int(uniq), C.byref(po), C.byref(n)))
        try:
            return (np.frombuffer((C.c_char * (4 * n.value)).from_address(po.value), dtype=np.uint32).copy()
                    if n.value else np.zeros(0, np.uint32))
        finally:
            gpu_lib().kssd_gpu_free(po)

    def set_filter(self, off, ids, pan, keep_members):
        """every sketch of the CSR without / restricted to the ids of pan, order kept -- kssd set -s / -i"""
        off = np.ascontiguousarray(off, dtype=np.uint64)
        ids = np.ascontiguousarray(ids, dtype=np.uint32)
        pan = np.ascontiguousarray(pan, dtype=np.uint32)
        n = len(off) - 1
        po, pi = C.c_void_p(), C.c_void_p()
        _gck(gpu_lib().kssd_gpu_set_filter(self.h, off.ctypes.data, ids.ctypes.data, n, pan.ctypes.data, len(pan),
                                           int(bool(keep_members)), C.byref(po), C.byref(pi)))
        try:
            ooff = np.frombuffer((C.c_char * (8 * (n + 1))).from_address(po.value), dtype=np.uint64).copy()
            tot = int(ooff[-1])
            oids = (np.frombuffer((C.c_char * (4 * tot)).from_address(pi.value), dtype=np.uint32).copy()
                    if tot else np.zeros(0, np.uint32))
        finally:
            gpu_lib().kssd_gpu_free(po)
            gpu_lib().kssd_gpu_free(pi)
        return ooff, oids

    def concat_units_device(self, d_off_all, d_ids_all, world, n_per_unit, cap, d_roff, d_rids, stream=None):
        """the all-gathered sketch units of `world` ranks -> one CSR (device tensors; nothing is synchronised)"""
        _gck(gpu_lib().kssd_gpu_concat_units_device(self.h, _ptr(d_off_all), _ptr(d_ids_all), world, n_per_unit, cap,
                                                    _ptr(d_roff), _ptr(d_rids), stream))

    @staticmethod
    def allgather_sketches(ctxs, d_off_l, d_ids_l, n_per_rank, unit_ids, d_roff, d_rids, streams=None):
        """the exchange inside one process: ctxs[i] / tensors[i] on device i (kssd_gpu_allgather_sketches, RCCL)"""
        n = len(ctxs)
        vp = C.c_void_p
        arr = lambda xs: (vp * n)(*[vp(_ptr(x)) for x in xs])
        hs = (vp * n)(*[c.h.value if isinstance(c.h, vp) else c.h for c in ctxs])
        st = (vp * n)(*[vp(s) for s in streams]) if streams is not None else None
        _gck(gpu_lib().kssd_gpu_allgather_sketches(hs, n, arr(d_off_l), arr(d_ids_l), n_per_rank, unit_ids, arr(d_roff), arr(d_rids), st))

    def index_set_filter(self, enable, skip_row_begin=0, skip_row_end=0):
        """negative filter in front of the index for searches whose rows mostly miss; rows [begin, end) bypass it"""
        _gck(gpu_lib().kssd_gpu_index_set_filter(self.h, int(bool(enable)), skip_row_begin, skip_row_end))

    def set_lds_sort_limit(self, max_tuples):
        """genomes staging more tuples than this take the global-memory dedup path (0 = default); results unchanged"""
        _gck(gpu_lib().kssd_gpu_set_lds_sort_limit(self.h, max_tuples))

    def set_fastq_quality(self, min_quality):
        """the quality floor (fastq2co -Q) of the following FASTQ text calls; 0 = none"""
        _gck(gpu_lib().kssd_gpu_set_fastq_quality(self.h, int(min_quality)))

    def fasta_read_starts(self, file=0):
        """read starts (u64 array) of FASTA file `file` of the batch tokenised last on this context (dist --byread)"""
        p, n = C.c_void_p(), C.c_uint64(0)
        _gck(gpu_lib().kssd_gpu_fasta_read_starts(self.h, file, C.byref(p), C.byref(n)))
        try:
            return np.frombuffer((C.c_char * (8 * n.value)).from_address(p.value), dtype=np.uint64).copy() if n.value else np.zeros(0, np.uint64)
        finally:
            if p.value:
                gpu_lib().kssd_gpu_free(p)

    def set_fastq_reads(self, on):
        """FASTQ text calls frame their input like dist -A (mt_shortreads2koc); what only the host does exactly is handed back"""
        _gck(gpu_lib().kssd_gpu_set_fastq_reads(self.h, int(bool(on))))

    def tuple_passes(self):
        """1, or 16 for k - drlevel = 9 (36-bit tuples: pass s keeps the tuples with low bits s, its ids are tuple >> 4)"""
        gpu_lib().kssd_gpu_tuple_passes.restype = C.c_uint32
        return int(gpu_lib().kssd_gpu_tuple_passes(self.h))

    def sketch_again(self, flags=SKETCH_FASTA, min_occ=1, with_pos=False):
        """the batch of the last host-level sketch call once more without a scan (the tuple passes 1 .. 15)"""
        n = C.c_uint32(0)
        po, pi, pp, bad = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_int64(-1)
        _gck(gpu_lib().kssd_gpu_sketch_again(self.h, flags, min_occ, C.byref(po), C.byref(pi), C.byref(pp) if with_pos else None, C.byref(bad)))
        try:
            ng = self._last_n
            off = np.frombuffer((C.c_char * (8 * (ng + 1))).from_address(po.value), dtype=np.uint64).copy()
            tot = int(off[-1])
            ids = np.frombuffer((C.c_char * (4 * tot)).from_address(pi.value), dtype=np.uint32).copy() if tot else np.zeros(0, np.uint32)
            pos = (np.frombuffer((C.c_char * (4 * tot)).from_address(pp.value), dtype=np.uint32).copy() if tot else np.zeros(0, np.uint32)) if with_pos else None
        finally:
            for q in (po, pi, pp):
                if q.value:
                    gpu_lib().kssd_gpu_free(q)
        return (off, ids, pos) if with_pos else (off, ids)

    def set_tuple_pass(self, s):
        _gck(gpu_lib().kssd_gpu_set_tuple_pass(self.h, int(s)))

    def set_scan_grid(self, max_workgroups):
        """at most this many scan workgroups (0 = one per CU): longer chunk runs per wave; results unchanged"""
        _gck(gpu_lib().kssd_gpu_set_scan_grid(self.h, max_workgroups))

    def scan_stats(self, stream=None):
        """(positions that passed the stage-1 filter, positions that also passed the Bloom test) of the last scan"""
        a, b = C.c_uint64(0), C.c_uint64(0)
        _gck(gpu_lib().kssd_gpu_scan_stats(self.h, C.byref(a), C.byref(b), stream))
        return a.value, b.value

    def set_kernel_timing(self, every):
        """bracket every `every`-th launch of the scan / rows kernels with events (1: all, 0: none)"""
        _gck(gpu_lib().kssd_gpu_set_kernel_timing(self.h, int(every)))

    def kernel_time(self, which, reset=False):
        """(average ms, launches) of the dominant kernel: 0 = sketch scan, 1 = distance rows"""
        ms, n = C.c_float(0), C.c_uint32(0)
        _gck(gpu_lib().kssd_gpu_kernel_time(self.h, which, int(reset), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def kernel_times(self, which):
        """the bracketed launches' durations one by one (ms, oldest first): 0 = sketch scan, 1 = distance rows"""
        buf = np.zeros(512, dtype=np.float32)
        n = C.c_uint32(0)
        _gck(gpu_lib().kssd_gpu_kernel_times(self.h, which, buf.ctypes.data, len(buf), C.byref(n)))
        return buf[:min(int(n.value), len(buf))].astype(np.float64)
