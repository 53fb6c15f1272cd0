#!/bin/bash
# A/B of library variants built under tools/_variants/ (liblvbgpu_<name>.so): the same probe with each, alternating, in one
# session.   gpurun -- bash tools/variant_ab.sh "<probe command>" <name> [<name> ...]     ("base" = the library as built)
probe=$1; shift
mkdir -p gpurun_out
cp lvb_amd/liblvbgpu.so /tmp/liblvbgpu_base.so
for rep in 1 2; do
  for v in "$@"; do
    if [ "$v" = base ]; then cp /tmp/liblvbgpu_base.so lvb_amd/liblvbgpu.so; else cp tools/_variants/liblvbgpu_$v.so lvb_amd/liblvbgpu.so; fi
    echo "== $v (run $rep)"
    timeout -k 10 300 $probe 2>&1 | grep -v "^\[anneal" | tail -n 6
    rc=${PIPESTATUS[0]}
    if [ $rc -eq 124 ] || [ $rc -ge 128 ]; then echo "stopping: $v was killed (rc $rc)"; cp /tmp/liblvbgpu_base.so lvb_amd/liblvbgpu.so; exit $rc; fi
  done
done
cp /tmp/liblvbgpu_base.so lvb_amd/liblvbgpu.so
