#!/usr/bin/env python
"""Turns a rocprofv3 kernel trace (CSV) into a small timeline: per 250 ms bin the summed run time of the front-end kernels, of the
feature-distance / hand-over kernels and the number of k_pair_loop dispatches alive, plus every k_pair_loop dispatch (start, end,
workgroups).  usage: rocprof_timeline.py <dir> <out.txt>"""
import csv
import glob
import os
import sys


def main():
    d, out = sys.argv[1], sys.argv[2]
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"],
                         int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0), int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1)))
    if not rows:
        open(out, "w").write("no kernel trace found\n")
        return
    t0 = min(r[0] for r in rows)
    loops = [(a - t0, b - t0, g // max(1, w)) for a, b, n, g, w in rows if "k_pair_loop" in n]
    span = max(r[1] for r in rows) - t0
    nb = int(span / 250e6) + 1
    fe, hand = [0.0] * nb, [0.0] * nb
    for a, b, n, g, w in rows:
        if "k_pair_loop" in n:
            continue
        tgt = hand if ("k_fd_" in n or "k_pairs_" in n or "k_transform" in n or "k_collect" in n) else fe
        x = a - t0
        while x < b - t0:
            i = int(x / 250e6)
            nxt = min(b - t0, (i + 1) * 250e6)
            tgt[i] += (nxt - x) / 1e6
            x = nxt
    lines = ["# bin start [s], front-end kernel ms in the bin (summed over streams), hand-over / feature-distance / S7 kernel ms, k_pair_loop dispatches alive"]
    for i in range(nb):
        mid = (i + 0.5) * 250e6
        alive = sum(1 for a, b, _ in loops if a <= mid < b)
        lines.append("%7.2f %9.1f %8.1f %3d" % (i * 0.25, fe[i], hand[i], alive))
    lines.append("# k_pair_loop dispatches: start [s], end [s], workgroups")
    for a, b, g in sorted(loops):
        lines.append("%8.3f %8.3f %6d" % (a / 1e9, b / 1e9, g))
    open(out, "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
