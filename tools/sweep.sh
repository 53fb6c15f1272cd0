#!/bin/bash
# tools/sweep.sh - run bench.py under a few tuning knobs and print the kernel time per launch
run() { python bench.py --steps 100 --warmup 10 --no-cpu-baseline --anneal-seconds 0 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['roofline']['launch_ms']*1e3,1),'us/launch', round(d['config']['batch']/d['roofline']['launch_ms']/1e3,1),'M/s kernel', round(d['value']/1e6,1),'M/s wall')"; }
for tw in 8192 16384 32768 65536 131072; do echo -n "TARGET_WAVES=$tw: "; LVBGPU_TARGET_WAVES=$tw run; done
for pad in 0 1 2 3 5; do echo -n "STRIDE_PAD_TILES=$pad: "; LVBGPU_STRIDE_PAD_TILES=$pad run; done
for B in 512 2048 8192; do echo -n "B=$B: "; run --batch $B; done
for mv in nni tbr; do echo -n "move=$mv: "; run --move $mv; done
