#!/usr/bin/env python3
"""Turn a gpurun_out/prof_<tag>_<mode>/ directory (written by tools/profile_gpu.sh on the MI355X box) into the
small, committed summaries under profiles/:

  profiles/<tag>_<mode>_graph_kernel_stats.csv   rocprofv3 --kernel-trace --stats of the default bench command
  profiles/<tag>_<mode>_eager_kernel_stats.csv   same, eager launches
  profiles/<tag>_<mode>_pmc_traffic.csv          per kernel: launches, FETCH_SIZE / WRITE_SIZE per launch (raw, KiB) and
                                                 the corrected HBM bytes per launch
  profiles/traffic.json                          {mode (or model_mode): {kernel name: bytes per launch}} read by bench.py

HBM-byte correction (MI355X_MICROARCH.md, "HBM"): on gfx950 FETCH_SIZE reports half of the bytes of a wide
coalesced streaming read, WRITE_SIZE is exact; both are in KiB.  traffic = (2*FETCH_SIZE + WRITE_SIZE) * 1024.
"""
import csv
import json
import shutil
import sys
from collections import defaultdict
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def pmc_per_kernel(path, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            tot[row["Kernel_Name"]] += float(row["Counter_Value"])
            cnt[row["Kernel_Name"]] += 1
    return tot, cnt


def main():
    tag, mode = sys.argv[1], sys.argv[2]
    src = ROOT / "gpurun_out" / f"prof_{tag}_{mode}"
    dst = ROOT / "profiles"
    dst.mkdir(exist_ok=True)
    for kind in ("graph", "eager"):
        f = src / kind / "trace_kernel_stats.csv"
        if f.exists():
            shutil.copy(f, dst / f"{tag}_{mode}_{kind}_kernel_stats.csv")
    for name in ("bench_graph.json", "bench_eager.json"):
        if (src / name).exists():
            shutil.copy(src / name, dst / f"{tag}_{mode}_{name}")
    ft, fc = pmc_per_kernel(src / "pmc_fetch" / "pmc_counter_collection.csv", "FETCH_SIZE")
    wt, wc = pmc_per_kernel(src / "pmc_write" / "pmc_counter_collection.csv", "WRITE_SIZE")
    rows = []
    for k in sorted(set(ft) | set(wt), key=lambda k: -(2 * ft.get(k, 0) + wt.get(k, 0))):
        n = max(fc.get(k, 0), wc.get(k, 0))
        fetch = ft.get(k, 0.0) / max(fc.get(k, 0), 1)
        write = wt.get(k, 0.0) / max(wc.get(k, 0), 1)
        rows.append((k, n, fetch, write, (2 * fetch + write) * 1024))
    with open(dst / f"{tag}_{mode}_pmc_traffic.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel_Name", "Launches", "FETCH_SIZE_KiB_per_launch_raw", "WRITE_SIZE_KiB_per_launch", "HBM_bytes_per_launch_corrected"])
        for r in rows:
            w.writerow([r[0], r[1], f"{r[2]:.1f}", f"{r[3]:.1f}", f"{r[4]:.0f}"])
    tj = dst / "traffic.json"
    data = json.loads(tj.read_text()) if tj.exists() else {}
    data[mode] = {"source": f"profiles/{tag}_{mode}_pmc_traffic.csv", "kernels": {r[0]: {"launches": r[1], "hbm_bytes_per_launch": round(r[4])} for r in rows}}
    tj.write_text(json.dumps(data, indent=1))
    for r in rows[:14]:
        print(f"{r[4] / 1e6:10.1f} MB/launch  x{r[1]:4d}  {r[0][:110]}")


if __name__ == "__main__":
    main()
