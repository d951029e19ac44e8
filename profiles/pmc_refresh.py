#!/usr/bin/env python3
"""Refresh profiles/pmc_traffic.json: the HBM traffic per launch of the hot kernels, from rocprofv3 --pmc passes.

Run it ON THE GPU BOX through gpurun, from the repository root, as `python3 profiles/pmc_refresh.py <tag>` -- it starts the
profiler itself, as child processes (never an exec of this process), with the program directly behind `--`:

    rocprofv3 --kernel-trace --pmc FETCH_SIZE  ... -- python3 bench.py --steps 2 --warmup 1 --cpu-sample 0 --spinup 0
    rocprofv3 --kernel-trace --pmc WRITE_SIZE  ... -- python3 bench.py ...            (separate passes: MI355X_MICROARCH.md)

and writes gpurun_out/pmc_traffic.json (copy it to profiles/ and commit it) with, per kernel, the per-launch averages of
FETCH_SIZE x 2 (the guide's gfx950 correction: 128-byte requests tallied at 64 -- calibrated on this repository's own
kernels in round 1, profiles/pmc_traffic.json history) and WRITE_SIZE, in bytes, and `source_sha` = the hash of the kernel
sources they were measured on (public_kssd_amd.capi.kernel_source_sha).  bench.py prints `traffic: null` when the
sources have changed since.
"""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
RX = "sketch_scan_kernel|sketch_dedup_kernel|sketch_gather_kernel|idx_|dist_rows_kernel|dist_metrics|tok_onepass_kernel|tok_summarise_kernel|tok_emit_kernel|mask_summarise_kernel"


def one_pass(tag, counter, out_dir):
    d = os.path.join(out_dir, "pmc_%s_%s" % (tag, counter))
    cmd = ["rocprofv3", "--kernel-trace", "--pmc", counter, "--kernel-include-regex", RX, "--output-format", "csv", "-d", d, "--",
           sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--cpu-sample", "0", "--spinup", "0"]
    env = dict(os.environ, TMPDIR="/tmp")
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    open(d + ".log", "wb").write(r.stdout)
    if r.returncode != 0:
        raise SystemExit("%s pass failed (rc %d): see %s.log" % (counter, r.returncode, d))
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not f:
        raise SystemExit("%s pass left no counter_collection.csv under %s" % (counter, d))
    acc, n = collections.defaultdict(float), collections.defaultdict(set)
    for row in csv.DictReader(open(f[0])):
        if row["Counter_Name"] != counter:
            continue
        k = row["Kernel_Name"].split("(")[0]
        acc[k] += float(row["Counter_Value"])
        n[k].add(row["Dispatch_Id"])
    return {k: acc[k] / len(n[k]) for k in acc}, {k: len(n[k]) for k in acc}


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    import public_kssd_amd as K
    fetch, nf = one_pass(tag, "FETCH_SIZE", out_dir)     # KB per dispatch
    write, nw = one_pass(tag, "WRITE_SIZE", out_dir)
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        kernels[k] = {"fetch_kb": fetch.get(k), "fetch_x2_bytes": 2.0 * 1024.0 * fetch.get(k, 0.0), "write_bytes": 1024.0 * write.get(k, 0.0),
                      "dispatches": [nf.get(k, 0), nw.get(k, 0)]}

    def total(prefix):
        ks = [k for k in kernels if prefix in k]
        return sum(kernels[k]["fetch_x2_bytes"] + kernels[k]["write_bytes"] for k in ks) if ks else None

    def fetch_only(prefix):
        ks = [k for k in kernels if prefix in k]
        return sum(kernels[k]["fetch_x2_bytes"] for k in ks) if ks else None

    res = {"tag": tag, "source_sha": K.capi.kernel_source_sha(),
           "source": "python3 profiles/pmc_refresh.py %s on the GPU box: two rocprofv3 --kernel-trace --pmc passes (FETCH_SIZE, WRITE_SIZE) of "
                     "bench.py --steps 2 --warmup 1 --cpu-sample 0 --spinup 0 (1 000 x 5 Mb), per-dispatch averages" % tag,
           "correction": 2.0,
           "correction_basis": "MI355X_MICROARCH.md section HBM: FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads on gfx950 "
                               "(128-byte requests tallied at 64); calibrated on this repository's kernels in round 1 (16, 8 and 4 bytes per "
                               "lane: 2^30 bytes read, 524 300 KB reported).  The x 2 is applied to narrow gathers too: an upper bound there. "
                               "WRITE_SIZE needs none (dist_rows_kernel: 36.0 MB written algorithmically, 35 210 KB reported).",
           # what bench.py prints as roofline.traffic: reads x 2 + writes of the scan (its writes are the candidate records: ~2 %)
           # (the headline's scan is the instantiation that reads the summary words, <SUBK, 0, 1>; the tokeniser's leg sketches once
           # without them: <SUBK, 0, 0> -- summed together before round 6's closing pass noticed)
           "sketch_scan_bytes_per_launch": total("sketch_scan_kernel<6, 0, 1>") if any("sketch_scan_kernel<6, 0, 1>" in k for k in kernels) else total("sketch_scan_kernel"),
           "sketch_scan_streamed_mask_bytes_per_launch": total("sketch_scan_kernel<6, 0, 0>"),
           # roofline_dist.traffic: what the rows kernel moves, reads AND writes (36 B per pair are writes)
           "dist_rows_bytes_per_launch": total("dist_rows_kernel"),
           # roofline_tok.traffic: the device tokeniser on the batch's genomes as FASTA text (reads of the text + the packed batch written)
           "tok_bytes_per_launch": total("tok_onepass_kernel"),
           "kernels": kernels}
    path = os.path.join(out_dir, "pmc_traffic.json")
    json.dump(res, open(path, "w"), indent=1)
    print(json.dumps({k: res[k] for k in ("tag", "source_sha", "sketch_scan_bytes_per_launch", "dist_rows_bytes_per_launch", "tok_bytes_per_launch")}))
    for k, v in kernels.items():
        print("%-60s fetch x2 %10.0f KB   write %10.0f KB   (%d / %d dispatches)" % (k[:60], v["fetch_x2_bytes"] / 1024, v["write_bytes"] / 1024,
                                                                                    v["dispatches"][0], v["dispatches"][1]))


if __name__ == "__main__":
    main()
