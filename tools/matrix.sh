#!/bin/bash
# tools/matrix.sh - the workloads of profiles/r0N_bench_matrix.txt, one line each (GPU box)
run() { python bench.py --steps 200 --warmup 20 --no-cpu-baseline --anneal-seconds 0 --no-shapes --no-configs --mixed-walk 0 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('| D %.2f | walk %.1f us | submit -> lengths %.1f M/s (%.1f us per step) | one at a time %.1f M/s | kernel only %.1f M/s | frac of L2 peak %.2f' % (d['config']['mean_dirty_nodes'], d['roofline']['launch_ms']*1e3, d['value']/1e6, d['ms_per_step']*1e3, d['one_at_a_time']['value']/1e6, d['kernel_only']['value']/1e6, d['roofline']['frac']))"; }
echo -n "cfg3 500x50k SPR B=4096 (default)            "; run
echo -n "cfg3 500x50k NNI B=4096                      "; run --move nni
echo -n "cfg3 500x50k TBR B=4096                      "; run --move tbr
echo -n "cfg3 500x50k SPR B=4096 after 3000 moves     "; run --walk 3000
echo -n "cfg3 500x50k SPR B=1024                      "; run --batch 1024
echo -n "cfg3 500x50k SPR B=16384                     "; run --batch 16384
echo -n "cfg2 64x10k NNI B=1024                       "; run --taxa 64 --sites 10000 --move nni --batch 1024
echo -n "cfg5 2000x200k TBR B=1024                    "; run --taxa 2000 --sites 200000 --move tbr --batch 1024
echo -n "cfg5 2000x200k TBR B=4096                    "; run --taxa 2000 --sites 200000 --move tbr --batch 4096
export LVBGPU_PAIR=2048
echo -n "cfg3 500x50k SPR B=4096, LVBGPU_PAIR=2048    "; run
echo -n "cfg3 500x50k SPR B=4096 +3000, LVBGPU_PAIR   "; run --walk 3000
