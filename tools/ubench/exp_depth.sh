for v in "" d1_2 d1_4 d2_3 d2_6 d7_4 d7_5 d7_6 d7_8; do
  if [ -z "$v" ]; then so=""; else so=zhilight_amd/build/variants/lib$v.so; fi
  echo "== variant ${v:-base}"
  ZHILIGHT_AMD_SO=$so python tools/bench_gemv.py --mfma --m 1 --layers 8 2>&1 | grep -v amdgpu.ids | head -8
done
