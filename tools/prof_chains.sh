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
# lanes: what does a post launch meet on the device?  For every post launch: was a scoring walk of ANOTHER queue running
# when it started, and how long did it take then
walks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"]) for r in rows if "fitch_walk<false" in r["Kernel_Name"]]
import bisect
ws = sorted(walks)
starts = [w[0] for w in ws]
alone, beside = [], []
for r in rows:
    if "post_kernel" not in r["Kernel_Name"]:
        continue
    st, en, q = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"]
    i = bisect.bisect_right(starts, st)
    other = any(w[2] != q and w[0] <= st < w[1] for w in ws[max(0, i - 4):i])
    (beside if other else alone).append(en - st)
if beside:
    print(f"\npost launches that started while another queue's walk ran: {len(beside)}, mean {sum(beside)/len(beside)/1000:.2f} us; "
          f"others: {len(alone)}, mean {sum(alone)/max(1,len(alone))/1000:.2f} us")
queues = sorted({r["Queue_Id"] for r in rows})
print("queues:", queues)
t0, t1 = int(rows[0]["Start_Timestamp"]), int(rows[-1]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
print(f"\nspan {(t1-t0)/1e6:.3f} ms, kernels busy {busy/1e6:.3f} ms ({100*busy/(t1-t0):.1f} %)")
PY
rm -rf $out
tail -3 gpurun_out/${tag}_chains${R}_run.log
cat gpurun_out/${tag}_chains${R}_trace_digest.txt
