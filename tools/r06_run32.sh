export TMPDIR=/tmp
mkdir -p gpurun_out/r06
for shp in "28672 4096 32" "28672 4096 1" "4096 4096 32"; do
  set -- $shp
  d=/tmp/ic_$1_$2_$3; rm -rf $d; mkdir -p $d
  timeout 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVES -d $d/pass0 -o p --output-format csv -- python tools/prof_one.py $1 $2 $3 6 mfma > $d/log 2>&1
  echo "== N=$1 K=$2 M=$3"; python tools/pmc_summary.py $d k_w4a16 2>&1 | tail -8
done > gpurun_out/r06/icache.txt 2>&1
cat gpurun_out/r06/icache.txt
