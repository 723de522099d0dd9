cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_w4.py -x -q -m gpu 2>&1 | tail -2
for m in 1 8 16 32; do timeout 100 python tools/bench_gemv.py --mfma --m $m 2>&1 | grep -E "plain|layer"; done
rm -rf gpurun_out/pmc_lds; timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS -d gpurun_out/pmc_lds/pass0 -o p --output-format csv -- python tools/prof_one.py 28672 4096 32 6 mfma > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/pmc_lds k_w4a16
