export TMPDIR=/tmp
mkdir -p gpurun_out/r06
for shp in "28672 4096 32" "4096 14336 32" "28672 4096 8"; do
  set -- $shp
  d=gpurun_out/r06/pmc_slab_$1_$2_m$3; rm -rf $d
  timeout 600 tools/pmc.sh $d k_w4a16_slab -- python tools/prof_one.py $1 $2 $3 6 mfma > /dev/null 2>&1
  echo "== k_w4a16_slab N=$1 K=$2 M=$3"; cat $d/summary.txt
done > gpurun_out/r06/pmc_slab.txt 2>&1
cat gpurun_out/r06/pmc_slab.txt
