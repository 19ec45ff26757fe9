#!/usr/bin/env python3
"""tests/golden/langevin_refdata.json from the reference's own known answers for its random engine and tabulated normal
distribution (src/gromacs/random/tests/refdata); inputs restated from random/tests/threefry.cpp:120-141 (zero, all-ones and
pi-digit keys / counters for ThreeFry2x64<0>) and tabulatednormaldistribution.cpp:57-70 (ThreeFry2x64<2>(123456, Other),
mean 2, stddev 5, ten 14-bit draws).  Data only; run in the build container."""
import json
import os
import xml.etree.ElementTree as ET

REF = "/root/reference/src/gromacs/random/tests/refdata"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "langevin_refdata.json")
INPUTS = [[0, 0, 0, 0], [0xFFFFFFFFFFFFFFFF] * 4,
          [0x243f6a8885a308d3, 0x13198a2e03707344, 0xa4093822299f31d0, 0x082efa98ec4e6c89]]   # ctr0, ctr1, key0, key1

out = dict(threefry=[], tabulated=None)
for i, inp in enumerate(INPUTS):
    seq = ET.parse(os.path.join(REF, "KnownAnswersTest_ThreeFry2x64Test_Default_%d.xml" % i)).getroot().find("Sequence")
    out["threefry"].append(dict(ctr=[str(v) for v in inp[:2]], key=[str(v) for v in inp[2:]], out=[u.text for u in seq.findall("UInt64")]))
seq = ET.parse(os.path.join(REF, "TabulatedNormalDistributionTest_Output14.xml")).getroot().find("Sequence")
out["tabulated"] = dict(key0=123456, domain=0, internalCounterBits=2, mean=2.0, stddev=5.0, values=[float(r.text) for r in seq.findall("Real")])
json.dump(out, open(OUT, "w"), indent=1)
print("wrote", OUT)
