cd $GRAFT_REPO_ROOT
o=gpurun_out/r05s; mkdir -p $o
for a in 1 2 3; do KSSD_DISTX_ABLATE=$a python3 profiles/rows_x_probe.py 2>&1 | grep "ablation\|rows_x" ; done | tee $o/rows_x_ablate.txt
