"""Timeline of one steady-state step of the domain-decomposition probe from a rocprofv3 kernel trace (kernel_trace.csv):
every kernel between two pack kernels with its start relative to the step's first kernel and its duration, averaged over the last
steps.  usage: dd_timeline.py <dir with *_kernel_trace.csv> [steps to average]"""
import csv
import glob
import os
import sys

d = sys.argv[1]
navg = int(sys.argv[2]) if len(sys.argv) > 2 else 50
f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def short(n):
    for a, b in (("void ", ""), ("(anonymous namespace)::", ""), ("nbnxm_hip::", "")):
        n = n.replace(a, b)
    return n.split("(")[0][:48]


# the first kernel of a step: the pack kernel (RCCL and peer-copy transports) or the coordinate-storing kernel (one-sided transport)
first = "haloPushCoordinatesKernel" if any("haloPushCoordinatesKernel" in r["Kernel_Name"] for r in rows) else "haloPackShifted"
starts = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
steps = []
for a, b in zip(starts[:-1], starts[1:]):
    steps.append(rows[a:b])
steps = steps[-navg - 1:-1]
# steps with the same kernel sequence as the last one
ref = [short(r["Kernel_Name"]) for r in steps[-1]]
same = [s for s in steps if [short(r["Kernel_Name"]) for r in s] == ref]
print("%d steps averaged (of %d), %d kernels per step" % (len(same), len(steps), len(ref)))
period = [int(b[0]["Start_Timestamp"]) - int(a[0]["Start_Timestamp"]) for a, b in zip(steps[:-1], steps[1:])]
print("step period (pack to pack) mean %.1f us" % (sum(period) / max(1, len(period)) / 1e3))
for k, name in enumerate(ref):
    st = sum(int(s[k]["Start_Timestamp"]) - int(s[0]["Start_Timestamp"]) for s in same) / len(same) / 1e3
    du = sum(int(s[k]["End_Timestamp"]) - int(s[k]["Start_Timestamp"]) for s in same) / len(same) / 1e3
    print("  %-48s start %7.1f us  duration %6.1f us  end %7.1f us" % (name, st, du, st + du))
