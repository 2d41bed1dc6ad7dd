#!/bin/bash
# tools/profile.sh -- rocprofv3 kernel trace + PMC passes of one bench workload.
#   bash tools/profile.sh TAG [CONFIG] [SHADING] [VARIANT]
# Run on the GPU box from the repo root; raw outputs under gpurun_out/prof_<TAG>/.  Each pass is a
# run of its own (a kernel trace with --stats, then four --pmc passes that stay within the
# per-block counter limits), every one under its own time limit; the script stops at the first
# pass that fails.  tools/pmc_summary.py turns the directory into the committed summaries.
set -u
TAG=${1:-r2}
CONFIG=${2:-c3_3840x2160_n1024}
SHADING=${3:-ref}
VARIANT=${4:-tiled}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 30 --warmup 5 --no-cpu --no-extras --config $CONFIG --shading $SHADING --variant $VARIANT"
echo "$CONFIG|$SHADING|$VARIANT" > "$OUT/workload.txt"
cd /tmp
timeout -k 10 180 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH > "$OUT/trace.log" 2>&1 || { echo "trace failed"; tail -5 "$OUT/trace.log"; exit 1; }
i=0
for PMC in \
  "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
  "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE" \
  "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
  "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum" ; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $PMC --output-format csv -d "$OUT/pmc$i" -- $BENCH > "$OUT/pmc$i.log" 2>&1 || { echo "pmc pass $i failed"; tail -5 "$OUT/pmc$i.log"; exit 1; }
done
echo "profile done: $OUT"
