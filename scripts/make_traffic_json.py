"""profiles/traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of `bench.py --steps 1 --warmup 1`.

usage: python scripts/make_traffic_json.py <fetch_counter_collection.csv> <write_counter_collection.csv> [<fetch2> <write2> ...]
HBM bytes per launch = f_read * FETCH_SIZE*1024 + f_write * WRITE_SIZE*1024, averaged over the launches of each kernel,
with PER-ACCESS-SHAPE correction factors from profiles/pmc_calibration.json (scripts/ubench/pmc_calib.hip: known-byte
kernels in the shapes the convolution kernels use; MI355X_MICROARCH.md section HBM documents only the 16-B-per-lane
shapes: FETCH_SIZE x2, WRITE_SIZE x1).  Which shape a kernel reads / writes in: `shapes_of` below."""
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


def load_calibration():
    """{shape: factor}; without a calibration file the documented factors of the 16-byte shapes are all there is"""
    path = os.path.join(ROOT, "profiles", "pmc_calibration.json")
    fac = {"read_b128_stream": 2.0, "read_lds_dma_b128": 2.0, "read_b64_patch": 2.0, "write_b128_stream": 1.0,
           "write_b32_stream": 1.0, "write_b32_rows": 1.0}
    src = "MI355X_MICROARCH.md defaults (no profiles/pmc_calibration.json)"
    if os.path.exists(path):
        cal = json.load(open(path))
        for side in ("fetch", "write"):
            for k, v in cal.get(side, {}).items():
                fac[k] = float(v["factor"])
        src = "profiles/pmc_calibration.json"
    return fac, src


def shapes_of(kernel: str):
    """(read shape, write shape) of a convolution kernel's dominant global accesses"""
    if kernel.startswith("conv3x3_wino_kernel"):       # 8-byte patch gathers (+ weight images by LDS-DMA), dword row stores
        return "read_b64_patch", "write_b32_rows"
    if kernel.startswith("conv3x3_wino_wgrad"):        # 8-byte patch / gradient gathers, dword stores of the dU blocks
        return "read_b64_patch", "write_b32_rows"
    if kernel.startswith(("conv3x3_bf16_dma", "conv3x3_wgrad_bf16_dma")):   # LDS-DMA reads; 16-byte staged stores / dword slabs
        return "read_lds_dma_b128", ("write_b32_rows" if "wgrad" in kernel else "write_b128_stream")
    if "bf16" in kernel:                               # register-staged bf16: 16-byte loads, 16-byte staged stores
        return "read_b128_stream", ("write_b32_rows" if "wgrad" in kernel else "write_b128_stream")
    return "read_b128_stream", "write_b32_rows"        # fp32 direct kernels: f32x4 loads, dword-per-lane 128-B row stores


def main():
    # one (fetch, write) pair per profiled command: fp32 first, then e.g. the bf16 leg's pair; `--out <file>` writes another
    # file than profiles/traffic.json (the inference leg's kernels partly carry the training kernels' names at other sizes)
    argv = list(sys.argv)
    out_name = "traffic.json"
    if "--out" in argv:
        i = argv.index("--out")
        out_name = argv[i + 1]
        del argv[i:i + 2]
    fetch, write = {}, {}
    for k in range(1, len(argv) - 1, 2):
        fetch.update(per_kernel(argv[k], "FETCH_SIZE"))
        write.update(per_kernel(argv[k + 1], "WRITE_SIZE"))
    out = {}
    fac, fac_src = load_calibration()
    for name, (v, k) in fetch.items():
        m = re.match(r"(?:void )?(conv[a-z0-9_]+kernel(?:<[^>]*>)?)", name)
        if not m:
            continue
        key = m.group(1)
        w = write.get(name, [0.0, 1])
        f_raw = v / k * 1024.0
        w_raw = w[0] / max(w[1], 1) * 1024.0
        rs, ws = shapes_of(key)
        out[key] = {"launches_profiled": k, "fetch_bytes_per_launch_raw": round(f_raw), "write_bytes_per_launch_raw": round(w_raw),
                    "read_shape": rs, "read_factor": fac[rs], "write_shape": ws, "write_factor": fac[ws],
                    "fetch_bytes_per_launch": round(fac[rs] * f_raw), "write_bytes_per_launch": round(fac[ws] * w_raw),
                    "hbm_bytes_per_launch": round(fac[rs] * f_raw + fac[ws] * w_raw)}
    out["_note"] = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `bench.py --steps 1 --warmup 1 "
                    "--no-cpu-baseline`; raw counters x per-access-shape factors (" + fac_src + "); values include Infinity-Cache "
                    "hits (the counters sit on the L2's fabric side)")
    json.dump(out, open(os.path.join(ROOT, "profiles", out_name), "w"), indent=1, sort_keys=True)
    for k in sorted(out):
        if k != "_note":
            print(k, out[k]["hbm_bytes_per_launch"])


if __name__ == "__main__":
    main()
