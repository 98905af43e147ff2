# What bounds the generated dK/dV loop?  Libraries with one ingredient removed (results are wrong; timing only).
# Build (in the container):  bash tools/probes/ablate_a64dkv.sh build      Run (GPU box): bash tools/probes/ablate_a64dkv.sh
C=csm-train-pytorch_amd/csrc
D=tools/probes/build/abl
if [ "$1" = build ]; then
  mkdir -p $D
  for b in 1 2 4 8 3 6 7; do
    CSM_A64DKV_ABLATE=$b python3 tools/gen/gen_attn64_dkv_loop.py $D/attn64_dkv_loop.inc &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -I$D -I$C -mllvm -amdgpu-spill-vgpr-to-agpr=0 -include $D/attn64_dkv_loop.inc -c $C/attention64_asm.hip -o $D/a64asm_$b.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libab_a64dkv_$b.so $(ls $C/build/*.o | grep -v attention64_asm) $D/a64asm_$b.o || exit 1
  done
  rm -f $D/*.o $D/attn64_dkv_loop.inc
  exit 0
fi
for lib in csm-train-pytorch_amd/csm/hip/libcsm_hip.so $D/libab_a64dkv_1.so $D/libab_a64dkv_2.so $D/libab_a64dkv_4.so $D/libab_a64dkv_8.so $D/libab_a64dkv_3.so $D/libab_a64dkv_6.so $D/libab_a64dkv_7.so; do
  echo "== $lib"
  CSM_HIP_LIB=$lib python3 tools/probes/a64_time2.py
done
