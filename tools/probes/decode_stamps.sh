# In-kernel timeline of the decode kernels (GPU box): builds a stamped copy of the library under /tmp and runs decode_stamps.py
cd $GRAFT_REPO_ROOT
C=csm-train-pytorch_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -DCSM_DECODE_STAMPS -c $C/generate.hip -o /tmp/generate_stamps.o &&
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libcsm_stamps.so $(ls $C/build/*.o | grep -v generate.hip) /tmp/generate_stamps.o &&
CSM_HIP_LIB=/tmp/libcsm_stamps.so python3 tools/probes/decode_stamps.py
