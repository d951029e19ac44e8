"""INTEGRATION.md's binding, compiled: tests/integration_stub/ref_gpu_seams.c holds the two functions a kssd maintainer
would add to the reference's command_dist.c; here it is checked against the REFERENCE's own headers (types, globals,
helper names) and this repository's two public headers.  Only where the reference sources exist (the dev container)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "command_dist.h")), reason="reference sources not present")
def test_the_reference_side_binding_compiles_against_the_reference_headers(tmp_path):
    src = os.path.join(ROOT, "tests", "integration_stub", "ref_gpu_seams.c")
    obj = str(tmp_path / "seams.o")
    r = subprocess.run(["gcc", "-std=gnu11", "-c", "-Wall", "-Werror=implicit-function-declaration", "-Werror=incompatible-pointer-types",
                        "-I" + REF, "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "public_kssd_amd", "host"),
                        src, "-o", obj], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode == 0, r.stdout.decode()
    # and every kssd_* symbol the object needs is one the two libraries export
    syms = subprocess.run(["nm", "-u", obj], stdout=subprocess.PIPE).stdout.decode().split()
    need = sorted(s for s in syms if s.startswith("kssd_"))
    have = set()
    for lib in ("libkssd_gpu.so", "libkssd_host.so"):
        out = subprocess.run(["nm", "-D", "--defined-only", os.path.join(ROOT, "public_kssd_amd", lib)], stdout=subprocess.PIPE).stdout.decode()
        have |= {line.split()[-1] for line in out.splitlines() if line.strip()}
    assert need and all(s in have for s in need), [s for s in need if s not in have]
