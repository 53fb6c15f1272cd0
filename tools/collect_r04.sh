#!/bin/bash
# tools/collect_r04.sh <suffix> - everything profiles/ quotes for one state of round 4, on the GPU box: kernel stats + PMC passes of the
# headline, of the mixed tree and of BASELINE's other shapes (profiles/collect.sh), the 32-chain and the one-chain annealing runs'
# kernels with a digest of their traces (tools/prof_chains.sh), and the post launch's roles (tools/post_profile.py).
S=${1:-a}
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
bash profiles/collect.sh r04${S} > gpurun_out/collect_r04${S}.log 2>&1; echo "headline done"
bash profiles/collect.sh r04${S}_mixed --walk 3075 > gpurun_out/collect_r04${S}_mixed.log 2>&1; echo "mixed done"
bash profiles/collect.sh r04${S}_cfg2 --taxa 64 --sites 10000 --move nni --batch 1024 > gpurun_out/collect_r04${S}_cfg2.log 2>&1; echo "cfg2 done"
bash profiles/collect.sh r04${S}_cfg5_b1024 --taxa 2000 --sites 200000 --move tbr --batch 1024 > gpurun_out/collect_r04${S}_cfg5_b1024.log 2>&1; echo "cfg5 b1024 done"
bash profiles/collect.sh r04${S}_cfg5_b4096 --taxa 2000 --sites 200000 --move tbr --batch 4096 > gpurun_out/collect_r04${S}_cfg5_b4096.log 2>&1; echo "cfg5 b4096 done"
bash tools/prof_chains.sh r04${S} 32 > gpurun_out/collect_r04${S}_chains32.log 2>&1; echo "chains32 done"
bash tools/prof_chains.sh r04${S} 1 > gpurun_out/collect_r04${S}_chains1.log 2>&1; echo "chains1 done"
( python3 tools/post_profile.py 1 12; python3 tools/post_profile.py 32 12 ) > gpurun_out/r04${S}_post_profile.txt 2>&1; echo "post profile done"
# what comes back is capped at 64 MiB: the summaries need the stats tables and the counter collections, not the traces
find gpurun_out/prof_r04${S}* -name "*_kernel_trace.csv" -delete
du -sh gpurun_out/prof_r04${S}*
