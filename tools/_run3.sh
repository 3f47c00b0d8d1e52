cd $GRAFT_REPO_ROOT
timeout 900 ./tools/bin/mb_tile_traffic
for u in 1024 2048 8192; do echo "=== 8x32 units $u"; ODIL_FUSED_UNITS=$u ODIL_HIP_LIB=$PWD/odil_amd/libodil_hip_8x32.so timeout 300 python3 tools/mb_adjoint_ab.py 2>&1 | grep "(512, 512, 512)"; done
