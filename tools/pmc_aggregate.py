#!/usr/bin/env python3
"""Fold rocprofv3 --pmc counter CSVs (one pass for FETCH_SIZE, one for WRITE_SIZE) into HBM bytes per launch and kernel.

    python tools/pmc_aggregate.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> N H L B

Units and corrections follow /opt/skills/guides (MI355X_MICROARCH.md, HBM section): both counters are in KiB; on gfx950
FETCH_SIZE reports half of the bytes of a wide coalesced streaming read, so bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.
"""
import csv
import json
import sys
from collections import defaultdict

CLASS_OF = (("ws_mlp_kernel", "wstat_mlp2"), ("w3_mlp_kernel", "wstat_mlp2"), ("w4_mlp_kernel", "wstat_mlp2"), ("w4_lin2_kernel", "wstat_sum2"),
            ("plane4_kernel", "plane_matmul"), ("ws_lin_kernel<2", "wstat_sum2"), ("w3_lin2_kernel", "wstat_sum2"),
            ("w3_lin_kernel<3, 2", "wstat_sum2"), ("w3_lin_kernel<1, 2", "wstat_sum2"), ("ws_lin_kernel<1", "wstat_cols"),
            ("w3_lin_kernel<3, 1", "wstat_cols"), ("w3_lin_kernel<1, 1", "wstat_cols"),
            ("plane_mm", "plane_matmul"), ("plane3_kernel", "plane_matmul"), ("plane3h_kernel", "plane_matmul"), ("head3_kernel", "head"), ("inorm_fwd", "inorm"), ("gemm_f32_kernel", "linear_gemm"))


def per_kernel(path, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            name = row["Kernel_Name"]
            tot[name] += float(row["Counter_Value"])
            cnt[name] += 1
    return {k: (tot[k] / cnt[k], cnt[k]) for k in tot}


def main():
    fetch_csv, write_csv, out, N, H, L, B = sys.argv[1:8]
    fetch = per_kernel(fetch_csv, "FETCH_SIZE")
    write = per_kernel(write_csv, "WRITE_SIZE")
    kernels = {}
    for name in sorted(set(fetch) | set(write)):
        f, nf = fetch.get(name, (0.0, 0))
        w, nw = write.get(name, (0.0, 0))
        kernels[name] = {"launches": max(nf, nw), "FETCH_SIZE_KiB_avg": f, "WRITE_SIZE_KiB_avg": w,
                         "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0}
    by_class = {}
    for cls in dict.fromkeys(c for _, c in CLASS_OF):
        sel = [(k, v) for k, v in kernels.items() if any(pat in k for pat, c in CLASS_OF if c == cls)]
        if not sel:
            continue
        n = sum(v["launches"] for _, v in sel)
        by_class[cls] = sum(v["hbm_bytes_per_launch"] * v["launches"] for _, v in sel) / max(n, 1)
    json.dump({"geometry": {"nodes": int(N), "hidden": int(H), "layers": int(L), "graphs_per_gpu": int(B)}, "precision": "bf16x3",
               "formula": "(2*FETCH_SIZE + WRITE_SIZE) * 1024 bytes, averaged over launches (gfx950 correction: FETCH_SIZE counts half of wide streaming reads)",
               "bytes_per_launch": by_class, "kernels": kernels}, open(out, "w"), indent=1)
    for k, v in by_class.items():
        print(f"{k:14s} {v / 1e6:10.1f} MB / launch")


if __name__ == "__main__":
    main()
