"""SQ / TCC / TCP counter breakdown per kernel from several rocprofv3 --pmc passes (one group of counters per pass).

    python tools/pmc_counters.py collect OUTDIR -- python tools/pmc_probe.py      (runs one rocprofv3 pass per group below)
    python tools/pmc_counters.py summarize OUTDIR [substring ...] > profiles/rNN_counters.txt

Every pass is `rocprofv3 --pmc <group> --kernel-trace --output-format csv` and nothing else (no --stats, no other trace
domain: the pool's gpurun refuses the combinations).  The summary prints per kernel the launches and the per-launch mean of
every counter (summed over the block instances), then the derived split the VERDICT asks for: share of the wave time parked
(SQ_WAIT_ANY), stalled at issue (SQ_WAIT_INST_ANY) and issuing (SQ_ACTIVE_INST_ANY), L2 hit rate, read requests per launch.
SQ_*CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md).
"""
import csv
import glob
import os
import subprocess
import sys
from collections import defaultdict

GROUPS = {
    "sq": "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS",
    "sq2": "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE",
    "sq3": "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES",
    "tcc": "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum",
    "tcc2": "TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_EA0_RDREQ_32B_sum",
    "tcc3": "TCC_TAG_STALL_sum TCC_BUSY_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum",
    "tcp": "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum",
    "grbm": "GRBM_GUI_ACTIVE GRBM_COUNT",
}


def collect(outdir, cmd, groups):
    os.makedirs(outdir, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    for g in groups:
        d = os.path.join(os.path.abspath(outdir), "pass_" + g)
        log = open(os.path.join(outdir, g + ".log"), "w")
        rc = subprocess.call(["timeout", "600", "rocprofv3", "--pmc"] + GROUPS[g].split() + ["--kernel-trace", "--output-format", "csv",
                             "-d", d, "-o", "p", "--"] + cmd, cwd="/tmp", env=env, stdout=log, stderr=subprocess.STDOUT)
        found = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if found:
            os.replace(found[0], os.path.join(outdir, g + "_counter_collection.csv"))
        subprocess.call(["rm", "-rf", d])
        print("pass", g, "rc", rc, "csv", bool(found), flush=True)


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]


def summarize(outdir, filters):
    per = defaultdict(lambda: defaultdict(list))  # kernel -> counter -> values per dispatch
    for path in sorted(glob.glob(os.path.join(outdir, "*_counter_collection.csv"))):
        by_dispatch = defaultdict(float)
        names = {}
        with open(path) as f:
            for r in csv.DictReader(f):
                key = (r["Dispatch_Id"], r["Counter_Name"])
                by_dispatch[key] += float(r["Counter_Value"])
                names[r["Dispatch_Id"]] = short(r["Kernel_Name"])
        for (disp, counter), v in by_dispatch.items():
            per[names[disp]][counter].append(v)
    for k in sorted(per):
        if filters and not any(s in k for s in filters):
            continue
        c = {name: sum(v) / len(v) for name, v in per[k].items()}
        launches = max(len(v) for v in per[k].values())
        print(f"[{k}]  launches per pass = {launches}")
        for name in sorted(c):
            print(f"      {name:34s} {c[name]:18.1f}")
        wc = c.get("SQ_WAVE_CYCLES")
        if wc:
            print("      -- wave time: parked (s_waitcnt / barrier) %.1f %%, issue-stalled %.1f %%, issuing %.1f %% (VALU %.1f %%, LDS %.1f %%)" % (
                100 * c.get("SQ_WAIT_ANY", 0) / wc, 100 * c.get("SQ_WAIT_INST_ANY", 0) / wc, 100 * c.get("SQ_ACTIVE_INST_ANY", 0) / wc,
                100 * c.get("SQ_ACTIVE_INST_VALU", 0) / wc, 100 * c.get("SQ_ACTIVE_INST_LDS", 0) / wc))
        if "TCC_HIT_sum" in c and c["TCC_HIT_sum"] + c.get("TCC_MISS_sum", 0) > 0:
            print("      -- L2: hit rate %.3f; EA read requests %.3e, write requests %.3e per launch" % (
                c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), c.get("TCC_EA0_RDREQ_sum", 0), c.get("TCC_EA0_WRREQ_sum", 0)))
        if "TCP_TCC_READ_REQ_LATENCY_sum" in c and c.get("TCP_TCC_READ_REQ_sum", 0) > 0:
            print("      -- L1->L2 read latency %.0f cycles per request" % (c["TCP_TCC_READ_REQ_LATENCY_sum"] / c["TCP_TCC_READ_REQ_sum"]))


if __name__ == "__main__":
    if sys.argv[1] == "collect":
        i = sys.argv.index("--")
        groups = [g for g in os.environ.get("PMC_GROUPS", ",".join(GROUPS)).split(",") if g in GROUPS]
        collect(sys.argv[2], sys.argv[i + 1:], groups)
    else:
        summarize(sys.argv[2], sys.argv[3:])
