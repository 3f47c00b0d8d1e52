cd $GRAFT_REPO_ROOT
for lib in libodil_hip_8x32.so libodil_hip_x_CHEAP_ADAM.so libodil_hip_x_NO_REDUCE.so libodil_hip_x_NO_G0.so libodil_hip_x_ALL.so libodil_hip_8x32.so; do
echo "=== $lib"; ODIL_HIP_LIB=$PWD/odil_amd/$lib timeout 300 python3 tools/mb_adjoint_one.py 2>&1 | grep full
done
