"""Mutated gzip streams through host/kssd_inflate.c (one at a time and two in step) under AddressSanitizer + UBSan: whatever the bytes, the
decoder returns the text zlib returns or refuses -- it never reads or writes outside its buffers.  CPU only:
  make -C public_kssd_amd asan && LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" \
  KSSD_HOST_LIB=$PWD/build/asan/libkssd_host.so ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 python profiles/fuzz_inflate.py 20000"""
import gzip, os, sys, zlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_inflate as T
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
def ref(z):
    """what zcat would write: every member, zero padding behind the last one accepted; None = refused"""
    out, rest = b"", z
    try:
        while rest:
            d = zlib.decompressobj(31)
            out += d.decompress(rest)
            if not d.eof: return None
            rest = d.unused_data
            if rest.strip(b"\0") == b"": break
        return out if z else None
    except zlib.error:
        return None
seeds = []
for lvl in (1, 6, 9):
    seeds.append(gzip.compress(T.dna(rng, int(rng.integers(100, 40_000))), lvl))
seeds.append(gzip.compress(T.dna(rng, 3000, alphabet=b"ACGTacgtNNNNRYKM"), 6) + gzip.compress(b"second member " * 50, 1))
co = zlib.compressobj(6, zlib.DEFLATED, 31, 8, zlib.Z_FIXED); seeds.append(co.compress(T.dna(rng, 5000)) + co.flush())
seeds.append(gzip.compress(rng.integers(0, 256, 3000, dtype=np.uint8).tobytes(), 6))
seeds.append(gzip.compress((b"AC" * 7 + b"GGT") * 900, 9))
for lvl in (1, 6, 9):  # text + noise + text in ONE stream: stored blocks right behind Huffman blocks, no flush between
    co = zlib.compressobj(lvl, zlib.DEFLATED, 31)
    seeds.append(co.compress(T.dna(rng, 30_000) + rng.integers(0, 256, 40_000, dtype=np.uint8).tobytes() + T.dna(rng, 9_000)
                             + rng.integers(0, 256, 700, dtype=np.uint8).tobytes() + T.dna(rng, 2_000)) + co.flush())
accepted = refused = differ = 0
prev = seeds[0]
for case in range(n_cases):
    z = bytearray(seeds[int(rng.integers(0, len(seeds)))])
    kind = int(rng.integers(0, 6))
    if kind == 0:   # bit flips
        for _ in range(int(rng.integers(1, 4))):
            z[int(rng.integers(0, len(z)))] ^= 1 << int(rng.integers(0, 8))
    elif kind == 1: # truncation
        z = z[: int(rng.integers(0, len(z)))]
    elif kind == 2: # a run of random bytes
        a = int(rng.integers(0, len(z))); b = min(len(z), a + int(rng.integers(1, 64)))
        z[a:b] = rng.integers(0, 256, b - a, dtype=np.uint8).tobytes()
    elif kind == 3: # bytes cut out of the middle
        a = int(rng.integers(10, len(z))); b = min(len(z), a + int(rng.integers(1, 32)))
        del z[a:b]
    elif kind == 4: # the trailer's length / CRC
        z[-int(rng.integers(1, 9))] ^= 0xFF
    z = bytes(z)
    want = ref(z)
    rc, got = T.gunzip(z)
    pair = T.gunzip2(z, prev) if case & 1 else T.gunzip2(prev, z)
    rc2, got2 = pair[0] if case & 1 else pair[1]
    if (rc == 0) != (rc2 == 0) or got != got2:
        print("one at a time and in step disagree: case", case); differ += 1
    if rc == 0:
        accepted += 1
        if want is None or got != want:
            # (zlib refuses what this decoder accepts only where gzip's optional header fields are damaged in ways the member's own
            # CRC-32 and length still vouch for the text)
            if want is not None: print("DIFFERENT TEXT: case", case); differ += 1
    else:
        refused += 1
        if want is not None: print("refused what zlib accepts: case", case, kind); differ += 1
    prev = z if len(z) < 200_000 else prev
print("cases %d accepted %d refused %d disagreements %d" % (n_cases, accepted, refused, differ))
sys.exit(1 if differ else 0)
