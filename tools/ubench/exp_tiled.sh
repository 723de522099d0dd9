# ablation of the M-tiled W4A16 GEMM (timing only; the ablated builds compute garbage): which of dequant VALU, LDS fragment
# reads, activation staging and the chunk barrier the 128 x 128 tile pays for.  usage (GPU box, repo root, after building
# the variants with tools/ubench/variant.sh t_<X> zhilight_amd/csrc/w4_gemm_tiled.hip -DZL_TEXP_<X>): bash tools/ubench/exp_tiled.sh
export ZL_W4_TILED_WIDE=-1
for v in full NODEQ NOLDS NOSTAGE NOBAR NODEQ_NOLDS ALL; do
  echo "== $v"
  if [ $v = full ]; then unset ZHILIGHT_AMD_SO; else export ZHILIGHT_AMD_SO=$PWD/zhilight_amd/build/variants/libt_$v.so; fi
  python tools/bench_gemv.py --mfma --m 1024 --iters 20 2>&1 | grep -v amdgpu.ids | tail -5
done
