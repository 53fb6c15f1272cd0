#!/usr/bin/env python3
"""Turn the raw rocprofv3 output of profiles/collect.sh (gpurun_out/prof_<tag>/) into the two
summaries kept under profiles/: <tag>_kernel_stats.csv (the --stats table as is) and
<tag>_pmc_summary.json (per-launch means of every collected counter for the scoring kernel, with the
HBM figures corrected as /opt/skills/guides/MI355X_MICROARCH.md prescribes), and refresh
profiles/traffic.json, which bench.py quotes as roofline.traffic.

    python profiles/summarize.py r01e
"""
import csv
import glob
import json
import shutil
import sys
from collections import defaultdict
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
KERNEL = "fitch_walk<false,"  # <COMMIT=false, WIDE=...>


def workload_of(extra) -> dict:
    """taxa / sites / batch / move of the profiled bench run, from the bench arguments it was taken with"""
    key = dict(taxa=500, sites=50000, batch=4096, move="spr")
    ex = list(extra)
    for i, a in enumerate(ex[:-1]):
        if a in ("--taxa", "--sites", "--batch"):
            key[a[2:]] = int(ex[i + 1])
        elif a == "--move":
            key["move"] = ex[i + 1]
    return key


def main(tag: str, mixed: bool = False, extra=()) -> None:
    if tag.startswith("-"):
        raise SystemExit(f"usage: summarize.py <tag> [mixed] [bench args ...]   ('{tag}' is not a tag)")
    src = ROOT / "gpurun_out" / f"prof_{tag}"
    if not src.is_dir():
        raise SystemExit(f"{src} does not exist: run profiles/collect.sh {tag} on the GPU box first")
    stats = glob.glob(str(src / "stats" / "*" / "*_kernel_stats.csv"))
    if stats:
        shutil.copy(stats[0], ROOT / "profiles" / f"{tag}_kernel_stats.csv")
    # <COMMIT = false, WIDE, HANDOVER>: the summary is of ONE variant of the scoring walk - the one the profiled run
    # launched most (the plain walk <.., 0> of the bench's pipelined loop; small shapes, whose steps all qualify as lone
    # steps, run the watcher variant <.., 2>); the handful of launches of the others are left out
    by_variant = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(str(src / "pmc_*" / "*" / "*_counter_collection.csv")):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                name = row["Kernel_Name"].split("(")[0]
                if KERNEL in name or "fitch_walk_pair<" in name:   # (the paired walk: A/B profiles taken with LVBGPU_PAIR)
                    by_variant[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
    variant = max(by_variant, key=lambda v: sum(len(x) for x in by_variant[v].values()), default=None)
    counters = by_variant[variant] if variant else {}
    if not counters and not stats:
        raise SystemExit(f"nothing under {src} matches the scoring kernel: no summary written")
    if not counters:
        print(f"no counter rows of the scoring kernel under {src}: kernel stats copied, no PMC summary written")
        return
    summary = {k: {"launches": len(v), "mean_per_launch": sum(v) / len(v)} for k, v in sorted(counters.items())}
    mean = lambda k: summary[k]["mean_per_launch"] if k in summary else None
    derived = {}
    if mean("FETCH_SIZE") is not None:
        # FETCH_SIZE counts KiB and sees exactly half of a 16-B/lane coalesced stream on gfx950: doubled
        derived["hbm_read_bytes_per_launch_corrected"] = mean("FETCH_SIZE") * 1024 * 2
    if mean("WRITE_SIZE") is not None:
        derived["hbm_write_bytes_per_launch"] = mean("WRITE_SIZE") * 1024
    if mean("TCC_HIT_sum") is not None and mean("TCC_MISS_sum") is not None:
        derived["l2_hit_rate"] = mean("TCC_HIT_sum") / (mean("TCC_HIT_sum") + mean("TCC_MISS_sum"))
    if mean("SQ_WAVES"):
        derived["waves_per_launch"] = mean("SQ_WAVES")
        for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_SMEM"):
            if mean(k) is not None:
                derived[k.lower().replace("sq_insts_", "") + "_insts_per_wave"] = mean(k) / mean("SQ_WAVES")
    if mean("SQ_WAVE_CYCLES"):
        derived["wave_cycle_shares"] = {n: mean(c) / mean("SQ_WAVE_CYCLES") for n, c in
                                        (("wait_memory(SQ_WAIT_ANY)", "SQ_WAIT_ANY"),
                                         ("wait_issue(SQ_WAIT_INST_ANY)", "SQ_WAIT_INST_ANY"),
                                         ("active(SQ_ACTIVE_INST_ANY)", "SQ_ACTIVE_INST_ANY")) if mean(c) is not None}
    doc = {"command": f"rocprofv3 --pmc <set> --output-format csv -- python3 bench.py --steps 40 --warmup 5 "
                      f"--headline-only {' '.join(extra)}  (one pass per counter set, profiles/collect.sh {tag})",
           "kernel": variant, "counters": summary, "derived": derived}
    (ROOT / "profiles" / f"{tag}_pmc_summary.json").write_text(json.dumps(doc, indent=1) + "\n")
    if "hbm_read_bytes_per_launch_corrected" in derived:
        # traffic.json: one entry per measured workload (bench.py picks the one that matches its arguments)
        tf = ROOT / "profiles" / "traffic.json"
        entries = json.loads(tf.read_text()) if tf.exists() else []
        if isinstance(entries, dict):
            entries = [entries]
        key = dict(workload_of(extra), mixed_walk=mixed)
        sha_file = src / "kernel_sha.txt"   # (profiles/collect.sh: the walk's sources at collection)
        sha = sha_file.read_text().strip() if sha_file.exists() else None
        entries = [e for e in entries if any(e.get(k, False if k == "mixed_walk" else None) != v for k, v in key.items())]
        entries.append(dict(key, hbm_bytes_per_launch=derived["hbm_read_bytes_per_launch_corrected"] + derived.get(
            "hbm_write_bytes_per_launch", 0.0), source=f"profiles/{tag}_pmc_summary.json", kernel_sha=sha))
        tf.write_text(json.dumps(entries, indent=1) + "\n")
    print(json.dumps(derived, indent=1))


if __name__ == "__main__":
    # python profiles/summarize.py <tag> [mixed] [bench args the profile was taken with ...]
    main(sys.argv[1] if len(sys.argv) > 1 else "r02a", "mixed" in sys.argv[2:3], [a for a in sys.argv[2:] if a != "mixed"])
