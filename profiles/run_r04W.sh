#!/bin/bash
# r04W: per-wave times of the scan on the bench batch (development build, KSSD_DEV_WAVETIME): how far the launch's end lies behind its average wave
tag=${1:-r04W}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
for i in 1 2; do KSSD_DEV_WAVETIME=1 timeout 300 profiles/scanbench 1000 5000000 10 | grep -v "^stats\|^whole"; done > gpurun_out/$tag/scan_wavetime.txt 2>&1
cat gpurun_out/$tag/scan_wavetime.txt
