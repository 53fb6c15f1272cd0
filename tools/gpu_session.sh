#!/bin/bash
# One gpurun call's worth of steps: each under its own timeout, logs under gpurun_out/; a step that is killed or times
# out ends the call (no GPU step is started behind a hung one), a step that merely fails does not.
#   gpurun -- bash tools/gpu_session.sh <tag> <step> [<step> ...]      steps: see the case below
tag=$1; shift
mkdir -p gpurun_out
step() { # name, seconds, command...
  local name=$1 secs=$2; shift 2
  echo "== $name" | tee -a gpurun_out/${tag}_session.log
  timeout -k 10 "$secs" "$@" > gpurun_out/${tag}_${name}.log 2>&1
  local rc=$?
  echo "== $name rc=$rc" | tee -a gpurun_out/${tag}_session.log
  tail -n 6 gpurun_out/${tag}_${name}.log
  if [ $rc -eq 124 ] || [ $rc -ge 128 ]; then echo "stopping: $name was killed (rc $rc)"; exit $rc; fi
}
for s in "$@"; do
  case $s in
    newtests) step newtests 600 python -m pytest tests/test_gpu_chains.py -x -q -m gpu -k "one_launch or oracle or decides or trajectory or runs_of or lanes" ;;
    pairs)    step pairs 500 python -m pytest tests/test_gpu_pairs.py tests/test_gpu_device_proposals.py -x -q -m gpu ;;
    walkab)   step walkab 300 python tools/walk_ab.py ;;
    walkabm)  step walkabm 300 python tools/walk_ab.py 4096 3075 ;;
    walkabm_p) LVBGPU_PAIR=2048 step walkabm_p 300 python tools/walk_ab.py 4096 3075 ;;
    walkab_p) LVBGPU_PAIR=2048 step walkab_p 300 python tools/walk_ab.py ;;
    chains)   step chains 400 python -m pytest tests/test_gpu_chains.py -x -q -m gpu ;;
    suite)    step suite 1000 python -m pytest tests -x -q -m gpu ;;
    probe1)   step probe1 200 python tools/chains_probe.py 1 ;;
    probe32)  step probe32 200 python tools/chains_probe.py 32 ;;
    probe32q) step probe32q 200 python tools/chains_probe.py 32 --quiet ;;
    probe32l1) PROBE_LANES=1 step probe32l1 200 python tools/chains_probe.py 32 ;;
    probe32l3) PROBE_LANES=3 step probe32l3 200 python tools/chains_probe.py 32 ;;
    probe32l4) PROBE_LANES=4 step probe32l4 200 python tools/chains_probe.py 32 ;;
    probe1r)  PROBE_RUN_LEVELS=3 step probe1r 200 python tools/chains_probe.py 1 ;;
    proposals) step proposals 700 python -m pytest tests/test_gpu_device_proposals.py tests/test_gpu_chains.py tests/test_gpu_pairs.py -x -q -m gpu ;;
    walkab1k) step walkab1k 300 python tools/walk_ab.py 1024 3075 ;;
    probe32q_p) LVBGPU_PAIR=1024 step probe32q_p 200 python tools/chains_probe.py 32 --quiet ;;
    probe1q_p) LVBGPU_PAIR=64 step probe1q_p 200 python tools/chains_probe.py 1 --quiet ;;
    probe32q_np) LVBGPU_PAIR=0 step probe32q_np 200 python tools/chains_probe.py 32 --quiet ;;
    probe1q)  step probe1q 200 python tools/chains_probe.py 1 --quiet ;;
    probe1q_np) LVBGPU_PAIR=0 step probe1q_np 200 python tools/chains_probe.py 1 --quiet ;;
    parity)   step parity 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu ;;
    pairq)    step pairq 400 python tools/pair_quality.py ;;
    pairqm)   step pairqm 400 python tools/pair_quality.py 4096 3075 ;;
    bench)    step bench 600 python bench.py ;;
    benchq)   step benchq 300 python bench.py --no-shapes --no-configs --no-cpu-baseline ;;
    *) echo "unknown step $s"; exit 2 ;;
  esac
done
