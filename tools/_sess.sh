cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05q
for r in 32 0 32; do echo "== PVD_INFER_VM_ROWS=$r"; PVD_INFER_VM_ROWS=$r PVD_HIP_LIB=$GRAFT_REPO_ROOT/aaai2023-pvd_amd/libpvd_hip_prof.so PVD_RENDER_KIND=vm PVD_RENDER_ONLY=p timeout 300 python tools/bench_render.py 2>&1 | grep -v amdgpu | grep "render\|persistent launch\|workgroup 0"; done | tee gpurun_out/r05q/render_vm_occ4.txt
