#!/usr/bin/env python3
"""Copy one scripts/profile_bench.sh run from gpurun_out/ into profiles/ and refresh traffic.json.

    python scripts/summarize_profiles.py <tag> <workload> <dominant-kernel-key> [round]

<tag> is the profile_bench.sh tag (gpurun_out/prof_<tag>), <workload> bench.py's --workload value,
<dominant-kernel-key> the key bench.py uses in kernels_ms ("decompress" | "compress").
HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (both in units of 1024 B as rocprofv3 reports
them; FETCH_SIZE doubled for wide coalesced reads, MI355X_MICROARCH.md HBM section).
"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL_OF = {"decompress": "k_decode_par<false>", "compress": "k_encode<"}


def find(d, suffix):
    hits = glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True)
    return max(hits, key=os.path.getmtime) if hits else None     # gpurun merges runs: take the newest


def pmc_avg(path, kernel_sub, counter):
    vals = []
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if kernel_sub in row["Kernel_Name"] and row["Counter_Name"] == counter:
                vals.append(float(row["Counter_Value"]))
    # the first launch of the workload is bench.py's setup/verification pass over the same data
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)


def main():
    tag, workload, key = sys.argv[1:4]
    rnd = sys.argv[4] if len(sys.argv) > 4 else "r03"
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    dst = os.path.join(ROOT, "profiles")
    base = "%s_bench_%s" % (rnd, workload)
    stats = find(os.path.join(src, "stats"), "kernel_stats.csv")
    shutil.copy(stats, os.path.join(dst, base + "_kernel_stats.csv"))
    for name in ("bench.json", "bench_under_rocprof.json"):
        p = os.path.join(src, name)
        if os.path.exists(p) and os.path.getsize(p):
            shutil.copy(p, os.path.join(dst, base + "_" + name))
    out = {}
    for leg, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        p = find(os.path.join(src, leg), "counter_collection.csv")
        # keep only our kernels' rows: the torch helper kernels are noise here
        with open(p, newline="") as f, open(os.path.join(dst, base + "_pmc_%s.csv" % leg), "w", newline="") as g:
            r = csv.reader(f)
            w = csv.writer(g, quoting=csv.QUOTE_NONNUMERIC)
            hdr = next(r)
            w.writerow(hdr)
            ki = hdr.index("Kernel_Name")
            for row in r:
                if row[ki].startswith("k_") or " k_" in row[ki]:
                    w.writerow(row)
        out[counter], n = pmc_avg(p, KERNEL_OF[key], counter)
        print(counter, out[counter], "KB avg over", n, "launches")
    traffic = int((2 * out["FETCH_SIZE"] + out["WRITE_SIZE"]) * 1024)
    tpath = os.path.join(dst, "traffic.json")
    t = json.load(open(tpath)) if os.path.exists(tpath) else {}
    t["%s:%s" % (workload, key)] = traffic
    notes = t.get("_notes", {})
    notes["%s:%s" % (workload, key)] = (
        "%s %s: FETCH_SIZE=%.4g KB (x2) + WRITE_SIZE=%.4g KB per launch of %s"
        % (rnd, tag, out["FETCH_SIZE"], out["WRITE_SIZE"], KERNEL_OF[key]))
    t["_notes"] = notes
    # every entry names the build it was measured at: the commit recorded next to the profile run (scripts/profile_bench.sh
    # writes gpurun_out/prof_<tag>/commit.txt on the GPU box from $PROFILE_COMMIT), else the checkout's HEAD
    commit = None
    cpath = os.path.join(src, "commit.txt")
    if os.path.exists(cpath):
        commit = open(cpath).read().strip() or None
    if not commit:
        try:
            import subprocess
            commit = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
        except Exception:
            commit = "unknown"
    commits = t.get("_commits", {})
    commits["%s:%s" % (workload, key)] = commit
    t["_commits"] = commits
    t.pop("_commit", None)
    json.dump(t, open(tpath, "w"), indent=1)
    print("traffic", traffic)
    with open(stats) as f:
        for line in f:
            if KERNEL_OF[key] in line:
                print(line.strip())


if __name__ == "__main__":
    main()
