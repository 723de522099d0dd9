#!/bin/bash
# PMC passes for one command (counters only: never combined with trace domains on this pool).
# usage: tools/pmc.sh <outdir> <kernel-substring> -- <command...>      (run from the repo root)
set -e
out=$1; pat=$2; shift 3
export TMPDIR=/tmp
mkdir -p "$out"
passes=(
 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY"
 "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT"
 "GRBM_GUI_ACTIVE TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum"
)
i=0
for p in "${passes[@]}"; do
  rocprofv3 --pmc $p -d "$out/pass$i" -o p --output-format csv -- "$@" > "$out/pass$i.log" 2>&1 || echo "pass $i failed (see $out/pass$i.log)"
  i=$((i+1))
done
python "$(dirname "$0")/pmc_summary.py" "$out" "$pat" | tee "$out/summary.txt"
