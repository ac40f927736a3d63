#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel (SQ counters are in quad-cycles)."""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
dur = collections.defaultdict(float)
seen = set()


def short(name):
    m = re.search(r"(\w+Kernel(?:<[^>]*>)?)", name)
    return m.group(1) if m else name[:40]


for r in rows:
    k = short(r["Kernel_Name"])
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Dispatch_Id"] not in seen:
        seen.add(r["Dispatch_Id"])
        cnt[k] += 1
        dur[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, v in sorted(agg.items(), key=lambda kv: -dur[kv[0]]):
    if cnt[k] < 3:
        continue
    line = "%-34s n=%5d avg_us=%8.1f" % (k, cnt[k], dur[k] / cnt[k] / 1e3)
    for name in sorted(v):
        line += " %s/disp=%.4g" % (name, v[name] / cnt[k])
    print(line)
    wc = v.get("SQ_WAVE_CYCLES")
    if wc:
        parts = []
        for name in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_LDS",
                     "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA"):
            if name in v:
                parts.append("%s=%.3f" % (name[3:], v[name] / wc))
        if "SQ_INSTS_VALU" in v and "SQ_WAVES" in v:
            parts.append("valu_insts/wave=%.0f" % (v["SQ_INSTS_VALU"] / max(v["SQ_WAVES"], 1)))
        if "SQ_WAVES" in v:
            parts.append("wavecycles/wave=%.0f(x4 cyc)" % (wc / max(v["SQ_WAVES"], 1)))
        if "GRBM_GUI_ACTIVE" in v and "SQ_INSTS_VALU" in v:
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs; a wave64 vector instruction holds its SIMD for 4 cycles
            # (1024 SIMDs on the chip)
            cycles = v["GRBM_GUI_ACTIVE"] / cnt[k] / 8.0
            secs = dur[k] / cnt[k] / 1e9
            parts.append("clock=%.2fGHz" % (cycles / secs / 1e9))
            parts.append("valu_issue_busy=%.2f" % (v["SQ_INSTS_VALU"] / cnt[k] / 1024.0 * 4.0 / cycles))
        print("      fraction of wave-cycles: " + " ".join(parts))
