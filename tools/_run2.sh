cd $GRAFT_REPO_ROOT
for lib in libodil_hip_base.so libodil_hip.so libodil_hip_8x32.so libodil_hip_4x64.so; do
echo "=== $lib"; ODIL_HIP_LIB=$PWD/odil_amd/$lib timeout 300 python3 tools/mb_adjoint_ab.py 2>&1 | grep TB
done
for lib in libodil_hip.so libodil_hip_8x32.so; do
echo "=== bench $lib"; ODIL_HIP_LIB=$PWD/odil_amd/$lib timeout 600 python3 bench.py 2>&1 | tail -1
done
ODIL_HIP_LIB=$PWD/odil_amd/libodil_hip_8x32.so timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_fullsize_gpu.py tests/test_properties_gpu.py -m gpu -x -q 2>&1 | tail -3
