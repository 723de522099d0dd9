for v in "" a5 a1 a2; do
  if [ -z "$v" ]; then so=""; else so=zhilight_amd/build/variants/lib$v.so; fi
  echo "== variant ${v:-full}"
  ZHILIGHT_AMD_SO=$so python tools/bench_gemv.py --mfma --m 1 --layers 8 2>&1 | grep -v amdgpu.ids | head -8
done
./tools/ubench/stream_probe | grep -E "phase order \+ 64|contiguous per wave"
