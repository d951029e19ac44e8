"""bench.py --gpus N without a launcher starts its own ranks (python -m torch.distributed.run, 127.0.0.1) before anything in
the parent process touches the GPU, and relays their exit code.  Here (no GPU) every rank stops with the bench's own
"needs a GPU" message: the launch path itself is what is checked; the N > 1 measurement runs on the GPU box."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_starts_its_own_ranks():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the N > 1 line itself is produced by profiles/run_r03a.sh")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    err = r.stderr.decode(errors="replace")
    assert r.returncode != 0                                   # the children's failure is the parent's
    assert "starting -m torch.distributed.run" in err and "--nproc-per-node 2" in err
    # a rank got as far as the bench's own check (the launcher ends the other ranks as soon as the first one has failed: whether
    # the second one's message still makes it out is a race)
    assert err.count("bench.py needs a GPU") >= 1, err[-2000:]
    assert r.stdout.decode().strip() == ""                      # no line without a measurement


def test_exchange_c_refuses_more_devices_than_there_are():
    """--exchange c drives the N devices from ONE process (no launcher, no torch.distributed): asked for more devices than the
    machine has it says so"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--exchange", "c", "--steps", "1", "--warmup", "0"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode != 0 and "--exchange c --gpus 64" in r.stderr.decode(errors="replace")
    assert r.stdout.decode().strip() == ""


@pytest.mark.gpu
def test_exchange_c_on_one_device_goes_through_rccl_and_holds_one_runtime():
    """bench.py --gpus 1 --exchange c: the whole step with kssd_gpu_allgather_sketches in it (a one-rank RCCL communicator on a
    one-GPU box), the line names the files the exchange ran on -- ONE HIP runtime and ONE RCCL mapped into the process"""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--exchange", "c", "--steps", "3", "--warmup", "1", "--spinup", "0",
                        "--genomes", "60", "--length", "400000", "--clades", "6"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode(errors="replace")[-3000:]
    j = json.loads([ln for ln in r.stdout.decode().splitlines() if ln.startswith('{"metric"')][-1])   # (RCCL prints its version banner to stdout too)
    assert j["n_gpus"] == 1 and j["exchange"]["kind"] == "c" and j["exchange"]["us"] > 0
    assert len(j["runtime"]["mapped"]["hip"]) == 1 and len(j["runtime"]["mapped"]["rccl"]) == 1
    assert os.path.dirname(j["runtime"]["rccl"]) == os.path.dirname(j["runtime"]["hip"])      # the RCCL next to the runtime in use
    assert j["matrix_checksum"] > 0 and j["value"] > 0


@pytest.mark.gpu
def test_two_ranks_share_one_device_over_gloo():
    """bench.py --gpus 2 on a one-GPU box: both ranks on cuda:0 (KSSD_BENCH_ONE_DEVICE), the exchange over gloo -- the whole N > 1 flow of
    the line (self-launch, process group, both partitions, max over ranks) except the transport"""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(KSSD_BENCH_ONE_DEVICE="1", KSSD_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--spinup", "0",
                        "--genomes", "60", "--length", "400000", "--clades", "6"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode(errors="replace")[-3000:]
    j = json.loads([ln for ln in r.stdout.decode().splitlines() if ln.startswith('{"metric"')][-1])
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["value"] > 0
    assert j["config"]["parallelism"]["ranks"] == 2 and j["config"]["parallelism"]["backend"] == "gloo"
    assert j["kernels"]["launches_timed"][0] >= 1
    # every rank's own terms side by side: what a measured scaling curve is read against (the one-GPU emulation prints the same terms)
    pr = j["per_rank"]
    assert [r["rank"] for r in pr] == [0, 1]
    for r in pr:
        assert r["step_ms"] > 0 and r["sketch_scan_ms"] > 0 and r["index_ms"] > 0 and r["rows_ms"] > 0 and r["exchange_us"] > 0 and r["ids"] > 0
    assert max(r["step_ms"] for r in pr) <= j["ms_per_step"] * 1.0001      # (the headline is the maximum over ranks)


def test_bench_modules_name_nothing_undefined():
    """bench.py and benchlib/ are only executed in full on a GPU box: every global name a function of theirs loads is defined in its
    module (an import, a definition, an assignment) -- what a split of the file can break without any CPU test noticing"""
    import ast
    import builtins
    for f in ("bench.py", os.path.join("benchlib", "workloads.py"), os.path.join("benchlib", "launch.py"), os.path.join("benchlib", "multi.py"),
              os.path.join("benchlib", "legs.py")):
        tree = ast.parse(open(os.path.join(ROOT, f)).read())
        defined = set(dir(builtins)) | {"__file__", "__name__"}
        for n in ast.walk(tree):
            if isinstance(n, (ast.FunctionDef, ast.ClassDef)):
                defined.add(n.name)
            elif isinstance(n, ast.Import):
                defined.update((a.asname or a.name).split(".")[0] for a in n.names)
            elif isinstance(n, ast.ImportFrom):
                defined.update(a.asname or a.name for a in n.names)
            elif isinstance(n, ast.Name) and isinstance(n.ctx, (ast.Store, ast.Del)):
                defined.add(n.id)
            elif isinstance(n, ast.arg):
                defined.add(n.arg)
            elif isinstance(n, ast.ExceptHandler) and n.name:
                defined.add(n.name)
        used = {n.id for n in ast.walk(tree) if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Load)}
        assert not (used - defined), (f, sorted(used - defined))


@pytest.mark.gpu
def test_one_gpu_plays_one_rank_of_three():
    """bench.py --emulate-world 3 --rank 1: the rank's whole step except xGMI (own batch sketched every step, the other ranks' units
    delivered by device-to-device copies, unpacking, index, the rank's rows) in BOTH partitions; the bench itself asserts that the
    own-index partition (transposed write) and the full index leave the same block, counts and metric bits"""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--emulate-world", "3", "--rank", "1", "--steps", "10", "--warmup", "1", "--spinup", "0",
                        "--genomes", "60", "--length", "400000", "--clades", "6", "--cpu-sample", "0"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode(errors="replace")[-3000:]
    j = json.loads([ln for ln in r.stdout.decode().splitlines() if ln.startswith('{"metric"')][-1])
    e = j["emulated"]
    assert j["n_gpus"] == 1 and e["world"] == 3 and e["rank"] == 1 and e["partition"] == "own"
    assert e["per_rank_ms"] > 0 and e["index_ms"] > 0 and e["rows_ms"] > 0 and e["exchange_bytes"] == 3 * (4 * j["exchange"]["unit_ids_per_rank"] + 8 * 61)
    assert e["partition_query"]["per_rank_ms"] > 0 and "bit-identical" in e["blocks_identical"]
    assert j["config"]["parallelism"]["ranks"] == 3 and "EMULATED" in j["config"]["parallelism"]["what"]
    assert j["kernels"]["launches_timed"][0] >= 1 and j["kernels"]["sketch_scan_spread"]["min_ms"] <= j["kernels"]["sketch_scan_ms"]
