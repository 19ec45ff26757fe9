"""Kernel timeline of a search step's first launch (tools/search_step_probe.py under rocprofv3 --kernel-trace): every kernel from the
first-pass prune kernel to the end of the force kernel that follows, with start offsets and gaps.  usage: search_step_timeline.py DIR"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
prunes = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("void nbnxmPruneKernel<true>")]
for p in prunes[1:]:
    t0 = int(rows[p]["Start_Timestamp"]); prev_end = t0
    print("--- search step")
    for r in rows[max(0, p - 6):p + 12]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print("  %+9.1f us  dur %7.1f  gap %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r["Kernel_Name"][:60]))
        prev_end = e
        if r["Kernel_Name"].startswith("void nbnxmKernel") and s > t0:
            break
