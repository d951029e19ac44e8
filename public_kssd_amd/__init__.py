"""public_kssd_amd -- MI355X (gfx950) implementation of the kssd sketch + distance hot path.

The product is native: HIP kernels behind a C ABI (include/kssd_gpu.h, csrc/), a C host layer and the
`kssd` command line (host/).  This package only binds those libraries for the tests and bench.py.
"""
from . import capi
from .capi import (Batch, GpuCtx, KssdError, Shuf, SketchSet, derive, distance_print, gpu_lib, host_lib, slot_order, slot_order_pos, slot_order_pos64, CHUNK_BASES, CHUNK_MASKW, CHUNK_WORDS,
                   SLACK_WORDS, SKETCH_FASTA, SKETCH_KEEP_ZERO, SKETCH_NO_CAPACITY, SKETCH_FIRST_POS, SKETCH_COUNTS, SKETCH_BY_POS, SKETCH_UNIQ, byread_write,
                   PHASE_PREP, PHASE_SCAN, PHASE_EXACT, PHASE_FINISH, PHASE_REPASS, device_count, dist_multi, distance_print_pairs)

__all__ = ["Batch", "GpuCtx", "KssdError", "Shuf", "SketchSet", "derive", "distance_print", "slot_order", "slot_order_pos", "slot_order_pos64", "byread_write",
           "gpu_lib", "host_lib", "CHUNK_BASES", "CHUNK_MASKW", "CHUNK_WORDS", "SLACK_WORDS", "SKETCH_FASTA", "SKETCH_KEEP_ZERO",
           "SKETCH_NO_CAPACITY", "SKETCH_FIRST_POS", "SKETCH_COUNTS", "SKETCH_BY_POS", "SKETCH_UNIQ", "PHASE_PREP", "PHASE_SCAN",
           "PHASE_EXACT", "PHASE_FINISH", "PHASE_REPASS", "device_count", "dist_multi", "distance_print_pairs"]
