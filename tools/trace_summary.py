#!/usr/bin/env python3
"""Per-grid summary of one kernel from a rocprofv3 kernel trace:  python3 tools/trace_summary.py <s_kernel_trace.csv> <name part>
rocprofv3's --stats averages every launch of a kernel name; the bench's `roofline` is about the launches of ONE size (the
outer proof's 2^21-row tables), so the two are compared per grid size."""
import collections
import csv
import sys


def main(path, part):
    groups = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if part in r["Kernel_Name"]:
            groups[(r["Kernel_Name"].split("(")[0], int(r["Grid_Size_X"]), int(r["Workgroup_Size_X"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    print("%-34s %10s %6s %6s %10s %10s %10s" % ("kernel", "grid", "block", "calls", "avg ms", "min ms", "max ms"))
    for (name, grid, block), v in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
        print("%-34s %10d %6d %6d %10.3f %10.3f %10.3f" % (name, grid, block, len(v), sum(v) / len(v), min(v), max(v)))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
