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


def test_command_restarts_itself_once_for_blocking_host_threads_and_never_under_preloaded_tooling():
    """host/kssd_cli.c main(): libgomp's threads wait by spinning unless OMP_WAIT_POLICY says otherwise when the library is LOADED, and
    spinning teams are what a container's CPU quota throttles -- the command starts itself again with the policy set, once, first thing.
    A preloaded library (a profiler's tool) may have started the GPU runtime already: such a process is never replaced."""
    import subprocess
    exe = os.path.join(ROOT, "public_kssd_amd", "kssd")
    base = {k: v for k, v in os.environ.items() if k not in ("OMP_WAIT_POLICY", "GOMP_SPINCOUNT", "KSSD_NO_REEXEC", "LD_PRELOAD") and not k.startswith(("ROCP", "HSA_TOOLS"))}   # (the GPU hosts preload a guard of their own: not part of this test)
    def starts(extra):
        r = subprocess.run([exe, "--version"], env=dict(base, OMP_DISPLAY_ENV="true", **extra), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
        assert r.returncode == 0 and r.stdout.strip()
        return r.stderr.decode().count("OPENMP DISPLAY ENVIRONMENT BEGIN")       # libgomp prints it once per process image
    assert starts({}) == 2
    assert starts({"KSSD_NO_REEXEC": "1"}) == 1
    assert starts({"OMP_WAIT_POLICY": "active"}) == 1                             # the caller's choice stands
    assert starts({"GOMP_SPINCOUNT": "1000"}) == 1
    assert starts({"ROCP_KSSD_TEST_MARK": "1"}) == 1                           # (any variable of the profilers' families: names of our own here)
    assert starts({"HSA_TOOLS_KSSD_TEST_MARK": "1"}) == 1
    libz = [p for p in ("/usr/lib/x86_64-linux-gnu/libz.so.1", "/lib/x86_64-linux-gnu/libz.so.1") if os.path.exists(p)]
    if libz:
        assert starts({"LD_PRELOAD": libz[0]}) == 2                               # a preloaded library that is no profiler's starts nothing
        import shutil, tempfile
        with tempfile.TemporaryDirectory() as td:                                 # ... one that carries a profiler's name does
            fake = os.path.join(td, "librocprofiler-sdk-tool.so")
            shutil.copy(libz[0], fake)
            assert starts({"LD_PRELOAD": fake}) == 1
