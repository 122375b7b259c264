#!/usr/bin/env python3
"""HBM traffic of the dominant kernel from rocprofv3 PMC passes (MI355X_MICROARCH.md, "HBM" + "rocprofv3 PMC slots").

FETCH_SIZE and WRITE_SIZE do not fit one pass (TCC slots), so they are collected in two runs of the same command:
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_FETCH_SIZE -o bench -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_WRITE_SIZE -o bench -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline
Units are KB. Calibration inside the very same runs, on kernels whose byte counts are known:
    WRITE_SIZE  k_fill<uint>      writes the 512 MiB word array once            -> factor = bytes / reported
    FETCH_SIZE  k_count_updated   reads two 512 MiB word arrays (+16 MiB bitmap) -> factor = bytes / reported
(the guide: on gfx950 FETCH_SIZE reports 1/2 of a coalesced streaming read; confirmed here for 4 B/lane loads).
Usage: python tools/pmc_traffic.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE k_relax_q > profiles/rNN_pmc_traffic.json
"""
import collections
import csv
import json
import os
import sys


def per_kernel(d, counter):
    acc = collections.defaultdict(list)
    with open(os.path.join(d, "bench_counter_collection.csv")) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]) * 1024.0)
    return acc


def find(acc, key):
    for k, v in acc.items():
        if key in k:
            return k, v
    raise KeyError(key)


def main():
    fdir, wdir, kern = sys.argv[1], sys.argv[2], sys.argv[3]
    grid = int(sys.argv[4]) if len(sys.argv) > 4 else 512
    F, W = per_kernel(fdir, "FETCH_SIZE"), per_kernel(wdir, "WRITE_SIZE")
    words = grid ** 3 * 4
    _, fill = find(W, "k_fill<unsigned int>")
    _, cnt = find(F, "k_count_updated")
    wfac = words / (sum(fill) / len(fill))
    ffac = (2 * words + grid ** 3 // 8) / (sum(cnt) / len(cnt))
    # every kernel whose name contains `kern` (the bulk path is several kernels per UpdateESDF: k_ft_rows, k_ft_plane<..>,
    # k_ft_x<..> and their empty overflow tiers); bytes are summed per UpdateESDF = per launch of k_ft_rows
    names = sorted(k for k in F if kern in k)
    if not names:
        raise KeyError(kern)
    first = [k for k in names if "k_ft_rows" in k or "k_nn_cells" in k]   # one launch of these per UpdateESDF
    n_updates = max(1, len(F[first[0]]) if first else len(F[names[0]]))
    fr_sum = sum(sum(F[k]) for k in names)
    wr_sum = sum(sum(W[k]) for k in names if k in W)
    kname = " + ".join(names)
    out = {
        "kernel": kname, "launches": n_updates,
        "fetch_reported_bytes_per_launch": fr_sum / n_updates, "write_reported_bytes_per_launch": wr_sum / n_updates,
        "fetch_calibration_factor": ffac, "write_calibration_factor": wfac,
        "calibration": {"write": "k_fill<uint> writes grid^3 x 4 B", "fetch": "k_count_updated reads 2 x grid^3 x 4 B + grid^3/8 B"},
        "fetch_bytes_per_launch": ffac * fr_sum / n_updates, "write_bytes_per_launch": wfac * wr_sum / n_updates,
        "per_kernel": {k: {"launches": len(F[k]), "fetch_bytes": ffac * sum(F[k]) / len(F[k]),
                           "write_bytes": wfac * sum(W[k]) / len(W[k]) if k in W else None} for k in names},
        "unit": "bytes per UpdateESDF (all its kernels)",
    }
    out["hbm_bytes_per_launch"] = out["fetch_bytes_per_launch"] + out["write_bytes_per_launch"]
    out["command"] = "python bench.py --steps 3 --warmup 1 --no-cpu-baseline (two rocprofv3 --pmc passes)"
    # the commit whose binary ran: the GPU box has no .git -- __graft_entry__.build() leaves a stamp that travels with the snapshot
    rev = os.environ.get("FIESTA_REV")
    if not rev:
        try:
            rev = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), ".fiesta_rev")).read().strip()
        except OSError:
            rev = "unrecorded"
    out["commit"] = rev
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
