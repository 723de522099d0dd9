set -x
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -f gpurun_out/r05_ab4.txt
for v in "" attn8occ4; do
  so=""; [ -n "$v" ] && so="$PWD/zhilight_amd/build/variants/lib$v.so"
  echo "== library ${v:-default (8-wave instantiation: 1 workgroup per CU)}" >> gpurun_out/r05_ab4.txt
  ZHILIGHT_AMD_SO=$so BATCHES=8 timeout 600 python tools/ab_step.py "ZL_ATTN_LA_SPLIT=128" "ZL_ATTN_LA_SPLIT=256" "ZL_ATTN_LA_SPLIT=288" "ZL_ATTN_LA_SPLIT=384,ZL_ATTN_LA_WAVES=4" "ZL_ATTN_LA_SPLIT=128" 2>&1 | grep -v amdgpu.ids >> gpurun_out/r05_ab4.txt
  ZHILIGHT_AMD_SO=$so BATCHES=16 timeout 600 python tools/ab_step.py "ZL_ATTN_LA_SPLIT=128" "base" "ZL_ATTN_LA_SPLIT=576,ZL_ATTN_LA_WAVES=4" "ZL_ATTN_LA_SPLIT=256,ZL_ATTN_LA_WAVES=4" "ZL_ATTN_LA_SPLIT=128" 2>&1 | grep -v amdgpu.ids >> gpurun_out/r05_ab4.txt
  ZHILIGHT_AMD_SO=$so BATCHES=32 timeout 600 python tools/ab_step.py "ZL_ATTN_LA_SPLIT=384,ZL_ATTN_LA_WAVES=4" "base" "ZL_ATTN_LA_SPLIT=1152,ZL_ATTN_LA_WAVES=4" "ZL_ATTN_LA=0" "ZL_ATTN_LA_SPLIT=384,ZL_ATTN_LA_WAVES=4" 2>&1 | grep -v amdgpu.ids >> gpurun_out/r05_ab4.txt
done
cat gpurun_out/r05_ab4.txt
