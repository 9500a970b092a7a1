#!/usr/bin/env python3
"""Per-kernel HBM traffic from the two PMC passes of profile_step.sh.

    python3 profiles/tools/summarize_traffic.py <dir> <tag>

Reads <dir>/<tag>_fetch/**/counter_collection.csv (FETCH_SIZE, KB) and <dir>/<tag>_write/** (WRITE_SIZE, KB), averages
per dispatch of every kernel whose name starts with one of the step's kernels, and writes <tag>_pmc_traffic.csv and
<tag>_pmc_traffic.json.  Corrections per MI355X_MICROARCH.md (HBM / rocprofv3 section): both counters are in KB;
gfx950 tallies a 128-byte read request as 64 B, so FETCH_SIZE is doubled (an upper bound for kernels whose reads are
64-byte requests); WRITE_SIZE is taken as is.
"""
import collections
import csv
import glob
import json
import re
import sys


def demangled(name: str) -> str:
    if name.startswith("_Z"):
        m = re.match(r"_Z(\d+)", name)
        if m:
            n = int(m.group(1))
            s = m.end()
            return name[s:s + n]
    return name.split("(")[0].split("<")[0].replace("void ", "").strip()


def per_kernel(dirpath: str, counter: str) -> dict:
    acc = collections.defaultdict(list)
    for f in glob.glob(f"{dirpath}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == counter:
                acc[demangled(row["Kernel_Name"])].append(float(row["Counter_Value"]))
    return acc


def main() -> None:
    d, tag = sys.argv[1], sys.argv[2]
    fetch = per_kernel(f"{d}/{tag}_fetch", "FETCH_SIZE")
    write = per_kernel(f"{d}/{tag}_write", "WRITE_SIZE")
    rows, js = [], {}
    total = 0.0
    for k in sorted(set(fetch) | set(write)):
        if k.startswith("__amd") or k.startswith("at::") or "elementwise" in k or "Cijk" in k:
            continue
        fv, wv = fetch.get(k, [0.0]), write.get(k, [0.0])
        # warm-up + timed steps launch each kernel once per step: the median is insensitive to the first cold launch
        fk = sorted(fv)[len(fv) // 2]
        wk = sorted(wv)[len(wv) // 2]
        fmb, wmb = 2.0 * fk * 1024 / 1e6, wk * 1024 / 1e6
        rows.append((k, len(fv), round(fk), round(fmb, 1), round(wk), round(wmb, 1), round(fmb + wmb, 1)))
        js[k] = {"launches": len(fv), "fetch_bytes": fmb * 1e6, "write_bytes": wmb * 1e6, "hbm_bytes": (fmb + wmb) * 1e6}
    # the step's own kernels launch once per step; one-off kernels (ring fill, prepare) stay out of the per-step sum
    most = max((r[1] for r in rows), default=0)
    total = sum(r[6] for r in rows if r[1] * 2 > most)
    with open(f"{d}/{tag}_pmc_traffic.csv", "w") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "launches", "FETCH_SIZE_raw_KB(median)", "fetch_corrected_MB (x2, gfx950)", "WRITE_SIZE_KB(median)",
                    "write_MB", "hbm_traffic_MB_per_launch"])
        w.writerows(rows)
        w.writerow(["sum_per_step", "", "", "", "", "", round(total, 1)])
    json.dump({"source": f"{tag}_pmc_traffic.csv", "unit": "bytes per launch (median over launches)",
               "corrections": "FETCH_SIZE KB x 1024 x 2 (gfx950 128-byte requests tallied as 64 B); WRITE_SIZE KB x 1024",
               "kernels": js, "sum_per_step_bytes": total * 1e6}, open(f"{d}/{tag}_pmc_traffic.json", "w"), indent=1)
    for r in rows:
        print(",".join(str(x) for x in r))
    print("sum_per_step_MB", round(total, 1))


if __name__ == "__main__":
    main()
