# Round profile: bench line, per-kernel stats, PMC of the dominant kernel, HBM traffic of the four decode GEMVs.
# usage (repo root, on the GPU box): bash tools/prof_round.sh r01       -> gpurun_out/<tag>_*
set -x
tag=${1:-r01}
export TMPDIR=/tmp
make -C oracle -s
timeout 900 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; tail -c 2500 gpurun_out/${tag}_bench.json
rm -rf gpurun_out/${tag}_prof; timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_prof -o bench --output-format csv -- python bench.py --no-cpu-baseline --steps 16 --warmup 2 > gpurun_out/${tag}_prof.log 2>&1
cp $(find gpurun_out/${tag}_prof -name 'bench_kernel_stats.csv' | head -1) gpurun_out/${tag}_bench_kernel_stats.csv
# the headline leg alone (batch 1 decode, no TTFT / batch legs): per-kernel averages of the decode step
rm -rf gpurun_out/${tag}_prof_b1; timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_prof_b1 -o bench --output-format csv -- python bench.py --no-cpu-baseline --no-ttft --no-extras --steps 32 --warmup 2 > gpurun_out/${tag}_prof_b1.log 2>&1
cp $(find gpurun_out/${tag}_prof_b1 -name 'bench_kernel_stats.csv' | head -1) gpurun_out/${tag}_decode_kernel_stats.csv
# counters in their own runs (no trace domains)
timeout 600 tools/pmc.sh gpurun_out/${tag}_pmc_gateup k_w4a16 -- python tools/prof_one.py 28672 4096 1 6 mfma > /dev/null; cp gpurun_out/${tag}_pmc_gateup/summary.txt gpurun_out/${tag}_gemv_gateup_pmc.txt
for shp in "6144 4096" "4096 4096" "28672 4096" "4096 14336"; do
  set -- $shp
  d=gpurun_out/${tag}_tcc_$1_$2; rm -rf $d; mkdir -p $d
  timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum -d $d/pass0 -o p --output-format csv -- python tools/prof_one.py $1 $2 1 6 mfma > $d/log 2>&1
  python tools/pmc_summary.py $d k_w4a16 > $d/summary.txt; cat $d/summary.txt
done
python - <<PY
import json, re
shapes = {"qkv 6144x4096": (6144, 4096), "o 4096x4096": (4096, 4096), "gate_up 28672x4096": (28672, 4096), "down 4096x14336": (4096, 14336)}
out = {"_what": "HBM read traffic per launch of the decode GEMVs from rocprofv3 --pmc TCC_EA0_RDREQ_sum (own pass, no tracing), x 128 B per request (gfx950 correction, MI355X_MICROARCH.md HBM); python tools/prof_one.py N K 1 6 mfma, average of 6 dispatches on cold (distinct) weights",
       "kernel": "k_w4a16_i8p (all four decode GEMVs at batch <= 4 since round 3; round 2: k_w4a16_phase / k_w4a16_mfma)", "shapes": {}}
tot = 0
for name, (n, k) in shapes.items():
    txt = open(f"gpurun_out/${tag}_tcc_{n}_{k}/summary.txt").read()
    rd = float(re.search(r"TCC_EA0_RDREQ_sum\s+([0-9.]+)", txt).group(1))
    out["shapes"][name] = {"rdreq": rd, "bytes": int(rd * 128)}
    tot += rd * 128
out["avg_bytes_per_launch"] = int(tot / 4)
json.dump(out, open("gpurun_out/${tag}_gemv_traffic.json", "w"), indent=2)
print(out)
PY
# prompt GEMM (gate|up at M = 1024): counters of the wide tile, own passes (PROF_PREFILL=1; the prompt kernels did not change in round 3)
[ -n "$PROF_PREFILL" ] && timeout 600 tools/pmc.sh gpurun_out/${tag}_pmc_wide k_w4a16_gemm_wide -- python tools/prof_one.py 28672 4096 1024 6 mfma > /dev/null; cp gpurun_out/${tag}_pmc_wide/summary.txt gpurun_out/${tag}_prefill_gemm_wide_gateup_m1024_pmc.txt
[ -n "$PROF_PREFILL" ] && python tools/bench_gemv.py --mfma --m 1024 --iters 20 2>&1 | grep -v amdgpu.ids | tail -5 > gpurun_out/${tag}_prefill_linears_m1024.txt
[ -n "$PROF_PREFILL" ] && python tools/bench_gemv.py --mfma --m 4096 --iters 10 2>&1 | grep -v amdgpu.ids | tail -5 > gpurun_out/${tag}_prefill_linears_m4096.txt
# (batch 8 / 32 and the INT8 route are legs of the default bench.py run since round 2: other_batches in the bench line)
timeout 300 python bench.py --no-cpu-baseline --no-ttft --no-extras --batch 32 --kv-cache-dtype int8 | tail -1 > gpurun_out/${tag}_bench_kvint8_b32.json 2>/dev/null
