cd $GRAFT_REPO_ROOT
for lib in "" F3 F4 F5; do
  if [ -n "$lib" ]; then export KSSD_GPU_LIB=$PWD/profiles/libkssd_gpu_$lib.so; else unset KSSD_GPU_LIB; fi
  python3 profiles/flat_probe.py 2>&1 | grep -v amdgpu.ids | grep "auto\|per row"
done | tee gpurun_out/r05h_filter_size_probe.txt
