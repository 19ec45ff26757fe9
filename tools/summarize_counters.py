#!/usr/bin/env python3
"""Merges the rocprofv3 counter passes of the force-only fused cluster kernel (tools/gpu_pmc.sh fused --primary-only and
tools/gpu_traffic.sh, each counter group in its own --pmc run as MI355X_MICROARCH.md prescribes) into ONE summary that
bench.py quotes with its provenance: profiles/r03/counters_fused_force_kernel.json.

usage: tools/summarize_counters.py <gpurun_out dir> <commit the profile was taken at> [out.json]"""
import collections
import csv
import glob
import json
import os
import re
import sys


def main():
    src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
    commit = sys.argv[2] if len(sys.argv) > 2 else "?"
    out = sys.argv[3] if len(sys.argv) > 3 else os.path.join("profiles", "r03", "counters_fused_force_kernel.json")
    flavour = sys.argv[4] if len(sys.argv) > 4 else "force"      # force: the force-only flavour; energy: the VF flavour (bench.py --timed-step energy)
    want = re.compile(r"nbnxmKernel<\d+, (false|true), \d+, %s, " % ("true" if flavour == "energy" else "false"))
    acc = collections.defaultdict(list)
    kernel = None
    for fn in glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv"), recursive=True) \
            + glob.glob(os.path.join(src, "traffic_*", "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(fn)):
            if want.search(row["Kernel_Name"]):
                kernel = row["Kernel_Name"][:80]
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    c = {k: sum(v) / len(v) for k, v in acc.items()}
    n = {k: len(v) for k, v in acc.items()}
    rec = {"kernel": kernel, "commit": commit, "launches_averaged": n, "counters": c,
           "how": "rocprofv3 --kernel-trace --pmc <one group per run> -- python3 bench.py --primary-only --no-cpu-baseline%s; averages per launch" % (" --timed-step energy" if flavour == "energy" else "")}
    g = c.get
    if g("SQ_INSTS_VALU") and g("SQ_BUSY_CYCLES"):
        # a wave64 VALU instruction occupies its SIMD for 2 cycles on gfx950 (MI355X_MICROARCH.md); 4 SIMDs x 256 CUs issue in parallel;
        # SQ_BUSY_CYCLES is summed over the 8 XCDs' SQs x 4 shader engines (32 per device on this part)
        if g("GRBM_GUI_ACTIVE"):
            cycles = g("GRBM_GUI_ACTIVE") / 8.0     # per XCD
            rec["valu_issue_frac"] = g("SQ_INSTS_VALU") * 2.0 / (cycles * 1024.0)
    if g("SQ_ACTIVE_INST_VALU") is not None and g("SQ_THREAD_CYCLES_VALU") and g("SQ_INSTS_VALU"):
        rec["active_lanes_per_valu_instruction"] = g("SQ_THREAD_CYCLES_VALU") / g("SQ_INSTS_VALU")
    if g("SQ_LDS_BANK_CONFLICT") is not None and g("SQ_LDS_IDX_ACTIVE"):
        rec["lds_bank_conflict_frac"] = g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE")
    if g("SQ_WAIT_ANY") and g("SQ_WAVE_CYCLES"):
        rec["wave_cycles_waiting_frac"] = g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES")
    if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
        # KiB; on gfx950 FETCH_SIZE reports half of a wide coalesced read stream (guide, HBM section): x2 = upper bound
        rec["hbm_fetch_bytes_raw"] = g("FETCH_SIZE") * 1024
        rec["hbm_write_bytes"] = g("WRITE_SIZE") * 1024
        rec["hbm_bytes_per_launch_corrected"] = 2 * g("FETCH_SIZE") * 1024 + g("WRITE_SIZE") * 1024
    os.makedirs(os.path.dirname(out), exist_ok=True)
    json.dump(rec, open(out, "w"), indent=1)
    print(json.dumps({k: v for k, v in rec.items() if k not in ("counters", "launches_averaged")}))


if __name__ == "__main__":
    main()
