#!/bin/bash
# anneal_chains with two candidates per wave from different batch sizes on (LVBGPU_PAIR=n; 0: never), alternating
for rep in 1 2; do
  for p in 0 512 1024 2048; do
    echo "== LVBGPU_PAIR=$p (run $rep)"
    LVBGPU_PAIR=$p timeout -k 10 200 python tools/chains_probe.py 32 16 --quiet 2>&1 | tail -n 2 | cut -c1-120
    rc=${PIPESTATUS[0]}
    if [ $rc -eq 124 ] || [ $rc -ge 128 ]; then echo "stopping (rc $rc)"; exit $rc; fi
  done
done
