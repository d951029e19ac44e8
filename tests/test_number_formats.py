"""The report's fast number formatters (host/kssd_formats.c kssd_fmt_f6 / kssd_fmt_e6) against the C library's own
"%.6lf" / "%E" -- the reference prints with snprintf (command_dist.c:1269-1286), so byte equality with the library is
byte equality with the reference."""
import ctypes as C
import struct

import numpy as np

import public_kssd_amd as K


def _fmt(fn, x):
    buf = C.create_string_buffer(400)
    end = fn(buf, C.c_double(x))
    return buf.raw[:end - C.addressof(buf)]


def _setup():
    L = K.host_lib()
    for f in (L.kssd_fmt_f6, L.kssd_fmt_e6):
        f.restype = C.c_void_p
        f.argtypes = [C.c_char_p, C.c_double]
    libc = C.CDLL(None)
    libc.snprintf.restype = C.c_int

    def want(fmt, x):
        b = C.create_string_buffer(400)
        n = libc.snprintf(b, C.c_size_t(400), fmt, C.c_double(x))
        return b.raw[:n]
    return L, want


def _values():
    rng = np.random.default_rng(7)
    v = [0.0, -0.0, 1.0, -1.0, 0.5, 2.0 ** -7, 3 * 2.0 ** -8, 0.0078125, 0.0234375, 1e-6, 5e-7, 4.9999999e-7, 1e-7, 9.9999995e-1, 0.9999995,
         0.99999949999, 9.9999995e6, 9999999.5, 10000005.0, 1e7, 1e22, 1e23, 123456789012.0, 5e11, 6e11, 1e300, 1e-300, 5e-324, 2.2250738585072014e-308,
         float("inf"), float("-inf"), float("nan"), -float("nan"), 1.7976931348623157e308, 0.1, 0.2, 0.3, 1 / 3, 2 / 3]
    v += list(rng.random(20000))                                   # what Jaccard / containment / distances look like
    v += list(rng.random(5000) * 1e-3) + list(-rng.random(2000))   # small metrics, negative CI bounds
    v += list(10.0 ** rng.uniform(-320, 308, 20000))               # p-values over the whole exponent range
    v += list(-(10.0 ** rng.uniform(-30, 12, 2000)))
    v += [struct.unpack("<d", struct.pack("<Q", int(b)))[0] for b in rng.integers(0, 2 ** 64, 20000, dtype=np.uint64)]  # any bit pattern
    # exact ties of "%.6f": (2N + 1) / 2^7 / 5^6 with 5^6 | 2N + 1
    v += [15625 * (2 * i + 1) / 2000000 for i in range(200)]
    # values next to the decade boundaries of "%E"
    for d in range(-20, 20):
        for eps in (-1e-9, -1e-12, 0, 1e-12, 1e-9):
            v.append(9.9999995 * 10.0 ** d * (1 + eps))
            v.append(1.0 * 10.0 ** d * (1 + eps))
    return v


def test_fixed_and_scientific_formats_are_the_c_librarys():
    L, want = _setup()
    for x in _values():
        assert _fmt(L.kssd_fmt_f6, x) == want(b"%.6lf", x), ("f6", x.hex() if x == x else "nan")
        assert _fmt(L.kssd_fmt_e6, x) == want(b"%E", x), ("e6", x.hex() if x == x else "nan")
