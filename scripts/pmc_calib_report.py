"""profiles/pmc_calibration.json from the two --pmc passes of scripts/ubench/pmc_calib (known-byte kernels).

usage: python scripts/pmc_calib_report.py <FETCH_SIZE counter_collection.csv> <WRITE_SIZE counter_collection.csv> [bytes_per_launch]
For every calibration kernel: reported = Counter_Value * 1024 B averaged over its launches; factor = true / reported —
the number a kernel's raw FETCH_SIZE / WRITE_SIZE is multiplied with when its accesses have that shape
(scripts/make_traffic_json.py)."""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = ["read_b128_stream", "read_b64_patch", "read_lds_dma_b128", "write_b128_stream", "write_b32_stream", "write_b32_rows"]


def per_kernel(path, counter):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            for n in NAMES:
                if n in r["Kernel_Name"]:
                    agg[n][0] += float(r["Counter_Value"])
                    agg[n][1] += 1
    return {k: v[0] / v[1] * 1024.0 for k, v in agg.items() if v[1]}


def main():
    true = float(sys.argv[3]) if len(sys.argv) > 3 else float(1 << 30)
    fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {"bytes_per_launch": true, "fetch": {}, "write": {},
           "_note": "scripts/ubench/pmc_calib.hip under rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes): every "
                    "kernel moves each byte of a 1 GiB buffer exactly once in the named access shape; factor = true bytes / "
                    "reported bytes (reported = counter * 1024)"}
    for n in NAMES:
        if n.startswith("read") and n in fetch:
            out["fetch"][n] = {"reported_bytes": round(fetch[n]), "factor": round(true / fetch[n], 4),
                               "write_side_reported_bytes": round(write.get(n, 0.0))}
        if n.startswith("write") and n in write:
            out["write"][n] = {"reported_bytes": round(write[n]), "factor": round(true / write[n], 4),
                               "fetch_side_reported_bytes": round(fetch.get(n, 0.0))}
    json.dump(out, open(os.path.join(ROOT, "profiles", "pmc_calibration.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
