"""Per-kernel HBM bytes per launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).

usage: python tools/pmc_summarize.py OUT/fetch_counter_collection.csv OUT/write_counter_collection.csv [n] [out.json]

With out.json, the corrected bytes per launch are also written as {"n": n, "kernels": {name: {"launches", "hbm_bytes"}}};
bench.py reads the newest profiles/*pmc_traffic.json of the matching n to fill `roofline.traffic`.

FETCH_SIZE / WRITE_SIZE are reported in KiB.  On gfx950 FETCH_SIZE counts 64 B per 128 B request for wide
coalesced reads (MI355X_MICROARCH.md §HBM), so the read side is calibrated on the k_scale launches of the
probe, whose traffic is known exactly (8n bytes read, 8n written): factor = known / measured, applied to
every kernel.  Both raw and corrected numbers are printed.
"""
import csv
import json
import sys
from collections import defaultdict


def load(path, counter):
    per = defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r.get("Counter_Name") != counter:
                continue
            per[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return per


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0][:60]


def summarize(fetch_csv, write_csv, n, verbose=False):
    """{"n", "calibration", "source", "kernels": {name: {"launches", "hbm_bytes"}}} from the two counter-collection CSVs."""
    fetch = load(fetch_csv, "FETCH_SIZE")
    write = load(write_csv, "WRITE_SIZE")
    known = 8.0 * n
    cal_r = cal_w = 1.0
    calibrated = False
    for k in fetch:
        if "k_scale" in k:
            cal_r = known / (1024.0 * sum(fetch[k]) / len(fetch[k]))
            calibrated = True
    for k in write:
        if "k_scale" in k:
            cal_w = known / (1024.0 * sum(write[k]) / len(write[k]))
    if verbose:
        print(f"calibration on k_scale (8n = {known:.3e} B each way): read x{cal_r:.3f}, write x{cal_w:.3f}")
        print(f"{'kernel':62s} {'launches':>8s} {'fetch_raw_MB':>13s} {'write_raw_MB':>13s} {'hbm_corrected_MB':>17s}")
    kernels = {}
    for k in sorted(fetch, key=lambda k: -sum(fetch[k])):
        fr = 1024.0 * sum(fetch[k]) / len(fetch[k])
        wr = 1024.0 * sum(write.get(k, [0.0])) / max(len(write.get(k, [0.0])), 1)
        if verbose:
            print(f"{short(k):62s} {len(fetch[k]):8d} {fr / 1e6:13.2f} {wr / 1e6:13.2f} {(fr * cal_r + wr * cal_w) / 1e6:17.2f}")
        kernels[short(k)] = {"launches": len(fetch[k]), "hbm_bytes": fr * cal_r + wr * cal_w}
    return {"n": n, "calibration": {"read": cal_r, "write": cal_w, "on": "k_scale (8n bytes each way)", "found": calibrated},
            "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on tools/pmc_probe.py", "kernels": kernels}


def main():
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 10_000_000
    d = summarize(sys.argv[1], sys.argv[2], n, verbose=True)
    if len(sys.argv) > 4:
        with open(sys.argv[4], "w") as f:
            json.dump(d, f, indent=1)


if __name__ == "__main__":
    main()
