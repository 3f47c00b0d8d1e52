cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_fullsize_gpu.py tests/test_properties_gpu.py tests/test_slab_gpu.py -m gpu -x -q 2>&1 | tail -3
for lib in libodil_hip_prev.so libodil_hip.so libodil_hip_prev.so libodil_hip.so; do
echo "=== $lib"; ODIL_HIP_LIB=$PWD/odil_amd/$lib timeout 300 python3 tools/mb_adjoint_one.py 2>&1 | grep full
done
