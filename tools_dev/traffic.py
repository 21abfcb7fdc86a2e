#!/usr/bin/env python3
"""dev: HBM traffic per launch of the bench's entry points from rocprofv3 PMC passes.

Run on the GPU box:  python3 tools_dev/traffic.py   (writes gpurun_out/traffic/, prints JSON)
FETCH_SIZE and WRITE_SIZE are collected in SEPARATE passes (they do not fit one pass, and the pool
refuses --pmc together with trace domains other than --kernel-trace).  Calibration: the same two
counters on 1 GiB streaming copies at 4 / 8 / 16 bytes per lane (tools_dev/fetch_calib.hip); the
per-width factors (true bytes / counter) are applied by each kernel's dominant access width."""
import csv, glob, json, os, subprocess, sys
root = os.environ.get("GRAFT_REPO_ROOT", os.getcwd())
out = os.path.join(root, "gpurun_out", "traffic")
os.makedirs(out, exist_ok=True)
env = dict(os.environ, TMPDIR="/tmp")

def pmc(tag, counter, cmd):
    d = os.path.join(out, f"{tag}_{counter}")
    subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "--"] + cmd,
                   cwd="/tmp", env=env, stdout=open(os.path.join(out, f"{tag}_{counter}.log"), "w"), stderr=subprocess.STDOUT)
    rows = {}
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                rows.setdefault(r["Kernel_Name"].split("(")[0], []).append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in rows.items()}

res = {"counters_unit": "KiB (FETCH_SIZE / WRITE_SIZE as reported)"}
GiB = float(1 << 30)
calib = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    m = pmc("calib", c, [os.path.join(root, "tools_dev", "fetch_calib")])
    for k, v in m.items():
        for wdt in ("4B", "8B", "16B"):
            if k.endswith("calib_copy_" + wdt):
                calib.setdefault(wdt, {})[c] = GiB / (v * 1024.0)   # true bytes per reported byte
res["calibration_true_over_reported"] = calib
steps, warm = 3, 1
bench = ["python3", os.path.join(root, "bench.py"), "--steps", str(steps), "--warmup", str(warm), "--no-cpu-baseline"]
raw = {c: pmc("bench", c, bench) for c in ("FETCH_SIZE", "WRITE_SIZE")}
res["raw_per_dispatch"] = {c: {k: v for k, v in m.items() if "waldo" in k} for c, m in raw.items()}
def corrected(kernel_sub, width):
    tot = 0.0
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for k, v in raw[c].items():
            if kernel_sub in k:
                tot += v * 1024.0 * calib.get(width, {}).get(c, 1.0)
    return tot
per = {
    "waldo_warp_composite_fwd": corrected("warp_composite_fwd_lds_kernel", "16B") + corrected("warp_composite_fwd_kernel", "4B"),
    "waldo_warp_composite_bwd": corrected("warp_composite_bwd_px16_kernel", "16B")
                                + corrected("warp_composite_splat_kernel", "16B") + corrected("gmap_reduce", "4B"),
}
per_kernel = {k.split("<")[0].split("::")[-1]: {c: raw[c].get(k, 0.0) * 1024.0 * calib.get("16B", {}).get(c, 1.0) for c in raw}
              for k in set(raw["FETCH_SIZE"]) | set(raw["WRITE_SIZE"]) if "waldo::warp" in k}
res["bytes_per_dispatch_by_kernel"] = per_kernel
res.update(frames=112, layers=8, height=256, width=512, bytes_per_launch=per,
           note="fwd = LDS-staged kernel (16-byte box loads); bwd = pixel kernel K1 (16-byte box loads, 16-byte "
                "record stores) + splat K2 (16-byte record loads) + partial reduce; workspace records written "
                "by K1 and re-read by K2 are real HBM traffic of this design")
json.dump(res, open(os.path.join(out, "traffic.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
