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


def test_command_runs_with_blocking_host_threads(tmp_path):
    """host/kssd_env.c + host/kssd_cli.c main(): libgomp's threads wait by spinning unless OMP_WAIT_POLICY says otherwise when the library
    INITIALISES, and spinning teams are what a container's CPU quota throttles.  libkssd_env.so -- linked behind libgomp, so initialised
    in front of it -- sets the passive policy from its constructor.  The command NEVER starts itself again (round 5's execv fallback is
    gone: a process that links the GPU runtime is not replaced): linked the other way round it runs on, spinning, as one process image."""
    import subprocess
    exe = os.path.join(ROOT, "public_kssd_amd", "kssd")
    base = {k: v for k, v in os.environ.items() if k not in ("OMP_WAIT_POLICY", "GOMP_SPINCOUNT", "OMP_THREAD_LIMIT", "LD_PRELOAD", "KSSD_TIMING")}
    def starts(extra, binary=exe):
        r = subprocess.run([binary, "--version"], env=dict(base, OMP_DISPLAY_ENV="verbose", **extra), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
        assert r.returncode == 0 and r.stdout.strip(), r.stderr.decode()[-400:]
        err = r.stderr.decode()
        blocks = err.split("OPENMP DISPLAY ENVIRONMENT BEGIN")[1:]                  # libgomp prints one per process image
        return len(blocks), "GOMP_SPINCOUNT = '0'" in blocks[-1], "wait_policy" in err
    assert starts({}) == (1, True, False)                                          # the constructor was in time: one image, no spinning
    assert starts({"KSSD_TIMING": "1"}) == (1, True, False)
    assert starts({"OMP_WAIT_POLICY": "active"}) == (1, False, False)              # the caller's choice stands
    n, passive, _ = starts({"GOMP_SPINCOUNT": "1000"}); assert n == 1 and not passive
    assert starts({"OMP_THREAD_LIMIT": "64"})[:2] == (1, True)                     # the caller's limit stands, the policy is set all the same
    # the same command line linked the other way round (libkssd_env in FRONT of libgomp): its constructor comes too late -- the command
    # goes on as it is, ONE process image, and says so where timing notes are asked for
    here = os.path.join(ROOT, "public_kssd_amd")
    late = str(tmp_path / "kssd_late")
    srcs = sorted(os.path.join(here, "host", f) for f in os.listdir(os.path.join(here, "host")) if f.startswith("kssd_cli") and f.endswith(".c"))
    r = subprocess.run(["gcc", "-std=gnu11", "-O1", "-fopenmp", "-fPIE"] + srcs + ["-o", late, "-I" + os.path.join(ROOT, "include"), "-L" + here, "-lkssd_host", "-lkssd_gpu",
                        "-Wl,-rpath," + here, "-Wl,-rpath,/opt/rocm/lib", "-lz", "-lm", "-lpthread", "-Wl,--no-as-needed", "-lkssd_env", "-lgomp"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert r.returncode == 0, r.stdout.decode()[-2000:]
    assert starts({}, late) == (1, False, False)
    assert starts({"KSSD_TIMING": "1"}, late) == (1, False, True)
    assert starts({"OMP_WAIT_POLICY": "passive"}, late) == (1, True, False)
    # no process-replacing call anywhere in the host sources
    for f in os.listdir(os.path.join(here, "host")):
        txt = open(os.path.join(here, "host", f)).read()
        assert "execv" not in txt and "execl" not in txt and "posix_spawn" not in txt and "fork(" not in txt, f
