#!/bin/bash
# r04l: fixed costs of a command: runtime start against the load of the library's code object (profiles/init_probe.hip)
tag=${1:-r04l}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
for i in 1 2 3; do ./profiles/init_probe; echo; done > gpurun_out/$tag/init_probe.txt 2>&1
cat gpurun_out/$tag/init_probe.txt
