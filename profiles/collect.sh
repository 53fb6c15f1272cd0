#!/bin/bash
# profiles/collect.sh - run on the GPU box (via gpurun) to collect the rocprofv3 evidence behind
# bench.py's roofline line.  Usage: bash profiles/collect.sh <round-tag> [bench args...]
# Writes gpurun_out/prof_<tag>/{stats,pmc_*}/...csv ; copy the summaries into profiles/ afterwards.
set -u
TAG=${1:-r01}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
ARGS="--steps 40 --warmup 5 --headline-only $*"
# what the counters are measured on: a hash of the scoring walk's sources (bench.py prints `stale` when they have moved)
( cd "$REPO" && python3 -c "import bench; print(bench.kernel_source_sha())" ) > "$OUT/kernel_sha.txt"
# pass 1: kernel trace + stats (durations)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$REPO/bench.py" $ARGS > "$OUT/stats.log" 2>&1
echo "stats rc=$?"
# PMC passes (counters only, no other tracing; TCC slots: FETCH_SIZE=3, WRITE_SIZE=2 -> separate passes)
i=0
for SET in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --output-format csv -d "$OUT/pmc_$i" -- python3 "$REPO/bench.py" $ARGS > "$OUT/pmc_$i.log" 2>&1
  echo "pmc_$i ($SET) rc=$?"
done
find "$OUT" -name "*.csv" | head -40
du -sh "$OUT"
