set -x
mkdir -p gpurun_out/r06
for m in 8 32; do
for v in "" nocomp nodeq nophase; do
  if [ -z "$v" ]; then so=""; else so=zhilight_amd/build/variants/lib$v.so; fi
  echo "== variant ${v:-full} M=$m"
  ZHILIGHT_AMD_SO=$so timeout 300 python tools/bench_gemv.py --mfma --m $m --layers 8 2>&1 | grep -v amdgpu.ids | head -12
done
done > gpurun_out/r06/ablate_rows.txt 2>&1
ZHILIGHT_AMD_SO=zhilight_amd/build/variants/libpprobe.so timeout 300 python tools/ubench/probe_phase_rows.py 8 32 > gpurun_out/r06/phase_timeline_rows.txt 2>&1
tail -50 gpurun_out/r06/ablate_rows.txt
