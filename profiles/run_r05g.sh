cd $GRAFT_REPO_ROOT
for lib in "" NT FLOOR; do
  if [ -n "$lib" ]; then export KSSD_GPU_LIB=$PWD/profiles/libkssd_gpu_$lib.so; else unset KSSD_GPU_LIB; fi
  python3 profiles/flat_probe.py 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r05g_flat_probe.txt
