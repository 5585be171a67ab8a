"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace CSV.

    python tools/trace_gaps.py OUT/*_kernel_trace.csv [first_kernel_substring]

Prints busy time, wall time and, per (previous kernel -> next kernel) pair, the number of transitions and the
mean / total gap — the cost of launch latency, event records and host round trips in the solver loop.
With a substring, everything before the first kernel whose name contains it is skipped (set-up phase).
"""
import csv
import sys
from collections import defaultdict


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "").replace("mispec::", "")
    return name.split("(")[0][:44]


def main():
    rows = []
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    if len(sys.argv) > 2:
        for i, r in enumerate(rows):
            if sys.argv[2] in r[2]:
                rows = rows[i:]
                break
    busy = sum(e - s for s, e, _ in rows)
    wall = rows[-1][1] - rows[0][0]
    print(f"kernels {len(rows)}  busy {busy / 1e6:.1f} ms  wall {wall / 1e6:.1f} ms  idle {(wall - busy) / 1e6:.1f} ms "
          f"({100.0 * (wall - busy) / wall:.1f} %)")
    pairs = defaultdict(lambda: [0, 0, []])
    for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
        g = max(0, s1 - e0)
        p = pairs[(n0, n1)]
        p[0] += 1
        p[1] += g
        p[2].append(g)
    print(f"{'previous -> next':92s} {'count':>7s} {'mean_us':>9s} {'total_ms':>9s} {'median':>8s} {'p90':>8s} {'max_us':>9s}")
    for (a, b), (c, t, gs) in sorted(pairs.items(), key=lambda kv: -kv[1][1])[:25]:
        gs.sort()
        print(f"{a + ' -> ' + b:92s} {c:7d} {t / c / 1e3:9.2f} {t / 1e6:9.2f} {gs[len(gs) // 2] / 1e3:8.2f} {gs[(9 * len(gs)) // 10] / 1e3:8.2f} "
              f"{gs[-1] / 1e3:9.2f}")


if __name__ == "__main__":
    main()
