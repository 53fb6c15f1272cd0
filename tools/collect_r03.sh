#!/bin/bash
# tools/collect_r03.sh <suffix> - everything profiles/ quotes for one state of round 3, on the GPU box: kernel stats + PMC passes of the
# headline, of the mixed tree and of BASELINE's other shapes (profiles/collect.sh), and the 32-chain annealing run's kernels.
S=${1:-a}
cd ${GRAFT_REPO_ROOT:-.}
bash profiles/collect.sh r03${S} > gpurun_out/collect_r03${S}.log 2>&1; echo "headline done"
bash profiles/collect.sh r03${S}_mixed --walk 3075 > gpurun_out/collect_r03${S}_mixed.log 2>&1; echo "mixed done"
bash profiles/collect.sh r03${S}_cfg2 --taxa 64 --sites 10000 --move nni --batch 1024 > gpurun_out/collect_r03${S}_cfg2.log 2>&1; echo "cfg2 done"
bash profiles/collect.sh r03${S}_cfg5_b1024 --taxa 2000 --sites 200000 --move tbr --batch 1024 > gpurun_out/collect_r03${S}_cfg5_b1024.log 2>&1; echo "cfg5 b1024 done"
bash profiles/collect.sh r03${S}_cfg5_b4096 --taxa 2000 --sites 200000 --move tbr --batch 4096 > gpurun_out/collect_r03${S}_cfg5_b4096.log 2>&1; echo "cfg5 b4096 done"
export TMPDIR=/tmp; R=$(pwd); cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r03${S}_chains32 -- python3 $R/tools/chains_probe.py --quiet 32 > $R/gpurun_out/collect_r03${S}_chains32.log 2>&1; echo "chains32 done"
cd $R
# what comes back is capped at 64 MiB: the summaries need the stats tables and the counter collections, not the traces
find gpurun_out/prof_r03${S}* -name "*_kernel_trace.csv" -delete
du -sh gpurun_out/prof_r03${S}*
