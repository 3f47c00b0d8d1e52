set -x
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_fullsize_gpu.py tests/test_properties_gpu.py -m gpu -x -q -k "adjoint or poisson or fused or transpose" 2>&1 | tail -5
echo "=== base"; ODIL_HIP_LIB=$PWD/odil_amd/libodil_hip_base.so timeout 300 python3 tools/mb_adjoint_ab.py
echo "=== new"; timeout 300 python3 tools/mb_adjoint_ab.py
echo "=== traffic"; timeout 600 ./tools/bin/mb_tile_traffic
