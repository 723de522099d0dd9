# ablation of k_w4a16_gemm_wide (timing only): usage as exp_tiled.sh, variants w_<X> built with -DZL_WEXP_<X>
for v in full NODEQ NOLDS NOSTAGE NOBAR ALL; do
  echo "== $v"
  if [ $v = full ]; then unset ZHILIGHT_AMD_SO; else export ZHILIGHT_AMD_SO=$PWD/zhilight_amd/build/variants/libw_$v.so; fi
  python tools/bench_gemv.py --mfma --m 1024 --iters 20 2>&1 | grep -v amdgpu.ids | grep "down\|gate_up"
done
