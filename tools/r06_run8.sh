mkdir -p gpurun_out/r06
for v in "" d12 d16; do
  if [ -z "$v" ]; then so=""; else so=zhilight_amd/build/variants/lib$v.so; fi
  echo "##### ring depth variant ${v:-d8}"
  ZHILIGHT_AMD_SO=$so timeout 900 python tools/bench_slab.py --m 32 16 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r06/slab_ring_depth.txt 2>&1
cat gpurun_out/r06/slab_ring_depth.txt
