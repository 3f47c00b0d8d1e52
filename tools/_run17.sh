cd $GRAFT_REPO_ROOT
timeout 600 python3 tools/slab_emulated.py 2 2>&1 | grep -v amdgpu
timeout 600 python3 tools/slab_emulated.py 3 2>&1 | grep -v amdgpu
