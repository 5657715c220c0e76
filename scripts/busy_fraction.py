#!/usr/bin/env python3
"""GPU busy fraction of a traced loop: sum of kernel durations / (end of the last kernel - start of the first) over the
steady-state window of a rocprofv3 --kernel-trace CSV (the first ``skip`` fraction of the kernels is dropped as warm-up).

    python scripts/busy_fraction.py <kernel_trace.csv> [skip=0.3]
"""
import csv
import sys


def main(path, skip=0.3):
    rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(path))))
    rows = rows[int(len(rows) * float(skip)):]
    busy, cur_end = 0, rows[0][0]
    for a, b in rows:                       # union of the intervals (kernels on several queues overlap)
        a = max(a, cur_end)
        if b > a:
            busy += b - a
            cur_end = b
    span = rows[-1][1] - rows[0][0]
    print(f"{len(rows)} kernels, window {span / 1e6:.2f} ms, busy {busy / 1e6:.2f} ms = {busy / span:.3f}")


if __name__ == "__main__":
    main(*sys.argv[1:3])
