"""profiles/traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of `bench.py --steps 1 --warmup 1`.

usage: python scripts/make_traffic_json.py <fetch_counter_collection.csv> <write_counter_collection.csv> [<fetch2> <write2> ...]
HBM bytes per launch = 2 * FETCH_SIZE*1024 (gfx950 counts 64 of every 128 B of a wide coalesced read:
MI355X_MICROARCH.md section HBM) + WRITE_SIZE*1024, averaged over the launches of each kernel."""
import collections
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_kernel(path, counter):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            a = agg[r["Kernel_Name"]]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
    return agg


def main():
    # one (fetch, write) pair per profiled command: fp32 first, then e.g. the bf16 leg's pair
    fetch, write = {}, {}
    for k in range(1, len(sys.argv) - 1, 2):
        fetch.update(per_kernel(sys.argv[k], "FETCH_SIZE"))
        write.update(per_kernel(sys.argv[k + 1], "WRITE_SIZE"))
    out = {}
    for name, (v, k) in fetch.items():
        m = re.match(r"(?:void )?(conv[a-z0-9_]+kernel(?:<[^>]*>)?)", name)
        if not m:
            continue
        key = m.group(1)
        w = write.get(name, [0.0, 1])
        f_raw = v / k * 1024.0
        out[key] = {"launches_profiled": k, "fetch_bytes_per_launch_raw": round(f_raw),
                    "fetch_bytes_per_launch_x2": round(2 * f_raw), "write_bytes_per_launch": round(w[0] / max(w[1], 1) * 1024.0),
                    "hbm_bytes_per_launch": round(2 * f_raw + w[0] / max(w[1], 1) * 1024.0)}
    out["_note"] = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `bench.py --steps 1 --warmup 1 "
                    "--no-cpu-baseline`; FETCH_SIZE doubled per MI355X_MICROARCH.md (wide coalesced reads are tallied at "
                    "64 of 128 B); values include Infinity-Cache hits (the counters sit on the L2's fabric side)")
    json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1, sort_keys=True)
    for k in sorted(out):
        if k != "_note":
            print(k, out[k]["hbm_bytes_per_launch"])


if __name__ == "__main__":
    main()
