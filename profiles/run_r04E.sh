#!/bin/bash
# r04E: the idle time between the step's six kernels (kernel trace with time stamps of the timed steps)
tag=${1:-r04E}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/$tag/prof -- python3 bench.py --steps 20 --warmup 5 --cpu-sample 0 > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/err.log
f=$(find gpurun_out/$tag/prof -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY' | tee gpurun_out/$tag/gaps.txt
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = ["sketch_scan", "sketch_dedup", "sketch_gather", "idx_scatter", "idx_build", "dist_rows"]
def short(n):
    for k in names:
        if k in n: return k
    return None
seq = [(short(r["Kernel_Name"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
seq = [x for x in seq if x[0]]
# whole steps: the six kernels in order
gaps = collections.defaultdict(list); durs = collections.defaultdict(list)
i = 0; steps = 0
while i + 6 < len(seq):
    if [x[0] for x in seq[i:i + 6]] == names and seq[i + 6][0] == "sketch_scan":
        for j in range(6):
            durs[names[j]].append(seq[i + j][2] - seq[i + j][1])
            nxt = seq[i + j + 1]
            gaps[names[j] + " -> " + nxt[0]].append(nxt[1] - seq[i + j][2])
        steps += 1; i += 6
    else:
        i += 1
print("whole steps found:", steps)
last = lambda v: v[-20:]
tot_d = tot_g = 0
for k in names:
    v = last(durs[k]); tot_d += sum(v) / len(v)
    print("  %-14s %8.2f us" % (k, sum(v) / len(v) / 1e3))
for k, v in gaps.items():
    v = last(v); tot_g += sum(v) / len(v)
    print("  gap %-32s %6.2f us (min %.2f max %.2f)" % (k, sum(v) / len(v) / 1e3, min(v) / 1e3, max(v) / 1e3))
print("  kernels %.2f us + gaps %.2f us = %.2f us per step (last 20 whole steps)" % (tot_d / 1e3, tot_g / 1e3, (tot_d + tot_g) / 1e3))
PY
rm -rf gpurun_out/$tag/prof
