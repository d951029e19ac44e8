#!/usr/bin/env python3
"""rocprofv3 --pmc passes of one bench.py command line for the kernels a regex names; per kernel and counter the per-dispatch average.
On the GPU box: python3 profiles/pmc_kernel.py <tag> <kernel regex> -- <bench.py arguments>"""
import collections, csv, glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, rx = sys.argv[1], sys.argv[2]
bench_args = sys.argv[sys.argv.index("--") + 1:]
PASSES = [["SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_LDS"],
          ["SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS", "GRBM_GUI_ACTIVE"],
          ["FETCH_SIZE"], ["WRITE_SIZE"], ["TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum"], ["TCP_TCC_READ_REQ_sum", "TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_ATOMIC_WITH_RET_REQ_sum"]]
out = os.path.join(ROOT, "gpurun_out")
res = collections.defaultdict(dict)
for pi, counters in enumerate(PASSES):
    d = os.path.join(out, "pmc_%s_%d" % (tag, pi))
    cmd = ["rocprofv3", "--kernel-trace", "--pmc"] + counters + ["--kernel-include-regex", rx, "--output-format", "csv", "-d", d, "--", sys.executable, os.path.join(ROOT, "bench.py")] + bench_args
    r = subprocess.run(cmd, cwd=ROOT, env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if r.returncode != 0 or not f:
        print("pass", counters, "failed rc", r.returncode, r.stdout.decode()[-400:]); continue
    acc, n = collections.defaultdict(float), collections.defaultdict(set)
    for row in csv.DictReader(open(f[0])):
        k = (row["Kernel_Name"].split("(")[0], row["Counter_Name"])
        acc[k] += float(row["Counter_Value"]); n[k].add(row["Dispatch_Id"])
    for (k, c), v in acc.items():
        res[k][c] = v / len(n[(k, c)]); res[k]["dispatches"] = len(n[(k, c)])
    subprocess.run(["rm", "-rf", d])
for k in sorted(res):
    print(k)
    for c, v in sorted(res[k].items()):
        print("    %-36s %.4g" % (c, v))
