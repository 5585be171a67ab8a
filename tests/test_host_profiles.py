"""The roofline fractions quoted for the two SpMV kernels, recomputed from the committed evidence:
rocprofv3 --kernel-trace --stats of C2-ONLY solves (profiles/r11f_c2_only_{dia,csr}_kernel_stats.csv — nothing else ran in those
processes, so the kernel's AverageNs is its C2 time), the HIP-event time of the same run (profiles/r11f_c2_only_*.json,
tools/c2_solves.py) and the PMC traffic of the same instantiations (profiles/r11f_c2_pmc_traffic.json).  No device needed."""
import csv
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PEAK = 8.0e12  # MI355X_MICROARCH.md: HBM3E ~ 8 TB/s
N, NNZ = 10_000_000, 149_595_984


def _stats(name):
    with open(os.path.join(ROOT, "profiles", name)) as f:
        return list(csv.DictReader(f))


def _kernel(rows, prefix):
    hits = [r for r in rows if prefix in r["Name"]]
    assert hits, prefix
    return max(hits, key=lambda r: int(r["Calls"]))  # the fused in-loop instantiation, not the bare stand-alone one


def _run(name):
    with open(os.path.join(ROOT, "profiles", name)) as f:
        return json.loads(f.read().strip().splitlines()[-1])


def test_headline_kernel_fraction_from_the_c2_only_trace():
    # diagonal storage: 15 diagonals of 8-byte values, x read once, y written once (first table row)
    bytes_dia = 8 * 15 * N + 8 * N + 8 * N
    k = _kernel(_stats("r11f_c2_only_dia_kernel_stats.csv"), "k_spmv_dia_win2<true")
    run = _run("r11f_c2_only_dia.json")
    assert run["format"] == 2 and run["num_operations"] == 1112 and run["num_iterations"] == 95 and run["nconv"] == 20
    assert int(k["Calls"]) >= 0.9 * run["num_operations"] * run["solves_traced"]  # (a sweep's first product: another instantiation)
    frac_trace = bytes_dia / (float(k["AverageNs"]) * 1e-9) / PEAK
    frac_events = bytes_dia / (run["spmv_ms_per_launch_hip_events"] * 1e-3) / PEAK
    assert 0.68 <= frac_trace <= 0.79, frac_trace          # 0.79 = the device's copy rate: nothing streams faster
    assert abs(frac_trace - frac_events) <= 0.05 * frac_trace, (frac_trace, frac_events)
    # the spread of the launches is small: MinNs / MaxNs of the one kernel of the one configuration
    assert float(k["MaxNs"]) <= 1.15 * float(k["MinNs"])


def test_csr_kernel_fraction_from_the_c2_only_trace():
    # 12 nnz + 20 n + 4 bytes per product for int32 CSR
    bytes_csr = 12 * NNZ + 20 * N + 4
    k = _kernel(_stats("r11f_c2_only_csr_kernel_stats.csv"), "k_spmv_csr_win<true")
    run = _run("r11f_c2_only_csr.json")
    assert run["format"] == 0 and run["csr_bytes_survey_8d"] == bytes_csr and run["num_operations"] == 1112
    frac_trace = bytes_csr / (float(k["AverageNs"]) * 1e-9) / PEAK
    frac_events = bytes_csr / (run["spmv_ms_per_launch_hip_events"] * 1e-3) / PEAK
    assert 0.62 <= frac_trace <= 0.79, frac_trace
    assert abs(frac_trace - frac_events) <= 0.05 * frac_trace, (frac_trace, frac_events)
    # same matrix, same solve: the two formats converge identically (bit-identical products)
    assert _run("r11f_c2_only_dia.json")["num_iterations"] == run["num_iterations"]


def test_pmc_traffic_is_close_to_the_algorithmic_bytes():
    with open(os.path.join(ROOT, "profiles", "r11f_c2_pmc_traffic.json")) as f:
        t = json.load(f)
    assert t["n"] == N and t["calibration"]["found"]
    kern = t["kernels"]
    dia = kern["k_spmv_dia_win2<true, 2, 6, false>"]["hbm_bytes"]
    csr = kern["k_spmv_csr_win<true, 1, 3, true, false>"]["hbm_bytes"]
    assert 1.0 <= dia / (8 * 15 * N + 16 * N) <= 1.03, dia       # no wasted re-reads
    assert 1.0 <= csr / (12 * NNZ + 20 * N + 4) <= 1.13, csr     # the windows' overlap re-reads x: 9 %
    # the one-sweep pass reads i + 4 vectors... its PMC traffic must be within 3 % of (columns + 4) * 8n (+ the written 2 * 8n)
    for cols8, name in ((9, "k_orth_lagged_dma<9, 3, true, true>"), (10, "k_orth_lagged_dma<10, 3, true, true>")):
        b = kern[name]["hbm_bytes"]
        assert 4 * (cols8 - 1) * 8 * N <= b <= 4 * cols8 * 8 * N + 6 * 8 * N, (name, b)
