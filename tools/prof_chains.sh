#!/bin/bash
# rocprofv3 kernel trace + stats of one lvbhost_anneal_chains run (R chains, 500 x 50k): gpurun_out/<tag>_chains<R>_*
#   gpurun -- bash tools/prof_chains.sh <tag> <R>
tag=$1; R=$2
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/${tag}_chains${R}
rm -rf $out && mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out -o trace --output-format csv -- python3 tools/chains_probe.py $R --quiet > gpurun_out/${tag}_chains${R}_run.log 2>&1
echo rc=$?
find $out -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/${tag}_chains${R}_kernel_stats.csv
# the trace itself is large: keep a digest - per kernel name count / mean / and the gaps between consecutive kernels
tr=$(find $out -name "*kernel_trace.csv" | head -1)
python3 - "$tr" > gpurun_out/${tag}_chains${R}_trace_digest.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: n.split("(")[0].replace("void lvbgpu::", "").replace("lvbgpu::", "")[:40]
dur = collections.defaultdict(list)
gap_after = collections.defaultdict(list)
for a, b in zip(rows, rows[1:]):
    dur[short(a["Kernel_Name"])].append(int(a["End_Timestamp"]) - int(a["Start_Timestamp"]))
    gap_after[(short(a["Kernel_Name"]), short(b["Kernel_Name"]))].append(int(b["Start_Timestamp"]) - int(a["End_Timestamp"]))
print("kernel                                    calls   mean us")
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k:40s} {len(v):6d} {sum(v)/len(v)/1000:9.2f}")
print("\ngap between consecutive kernels (end -> start), by pair: count, mean us, median us")
for k, v in sorted(gap_after.items(), key=lambda kv: -sum(kv[1]))[:14]:
    v2 = sorted(v)
    print(f"{k[0]:40s} -> {k[1]:40s} {len(v):6d} {sum(v)/len(v)/1000:9.2f} {v2[len(v2)//2]/1000:9.2f}")
t0, t1 = int(rows[0]["Start_Timestamp"]), int(rows[-1]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
print(f"\nspan {(t1-t0)/1e6:.3f} ms, kernels busy {busy/1e6:.3f} ms ({100*busy/(t1-t0):.1f} %)")
PY
rm -rf $out
tail -3 gpurun_out/${tag}_chains${R}_run.log
cat gpurun_out/${tag}_chains${R}_trace_digest.txt
