#!/usr/bin/env python3
"""Merge a FETCH_SIZE pass and a WRITE_SIZE pass of rocprofv3 --pmc into HBM bytes per kernel launch.

Both counters are in KiB.  On gfx950 FETCH_SIZE tallies the 128-byte requests of a wide coalesced read
(16 bytes per lane) at 64 bytes, so the read side is doubled; WRITE_SIZE is exact for 16-byte-per-lane
streaming stores.  Narrow scattered gathers are not calibrated, so for a gather-heavy kernel the read side
is an upper bound.
"""
import collections
import csv
import json
import re
import sys


def per_kernel(path, counter):
    tot = collections.defaultdict(float)
    n = collections.Counter()
    dur = collections.defaultdict(float)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        m = re.search(r"(\w+Kernel)", r["Kernel_Name"])
        if not m:
            continue
        k = m.group(1)
        tot[k] += float(r["Counter_Value"])
        n[k] += 1
        dur[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return {k: (tot[k] / n[k], n[k], dur[k] / n[k] / 1e3) for k in tot}


def main():
    workload, fetch_csv, write_csv = sys.argv[1:4]
    f = per_kernel(fetch_csv, "FETCH_SIZE")
    w = per_kernel(write_csv, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(f) | set(w)):
        fk, n, us = f.get(k, (0.0, 0, 0.0))
        wk = w.get(k, (0.0, 0, 0.0))[0]
        kernels[k] = {"launches": n, "avg_us_under_pmc": us, "fetch_size_kib": fk, "write_size_kib": wk,
                      "hbm_bytes_per_launch": (2.0 * fk + wk) * 1024.0}
    print(json.dumps({"workload": workload, "how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs of "
                      "bench.py --steps 1 --warmup 0 --pmc-pass (the launches of one timed step, nothing else); bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 per launch",
                      "kernels": kernels}, indent=1))


main()
