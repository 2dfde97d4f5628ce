#!/usr/bin/env python3
"""Sum kernel-census files ($QDAS_KERNEL_CENSUS=<file>, csrc/das_tile.hip tile_census) and print the prebuilt table of das_tile_cfg.h.

    QDAS_KERNEL_CENSUS=$PWD/gpurun_out/census/suite.txt python -m pytest tests -m gpu
    tools/kernel_census.py gpurun_out/census/*.txt          -> the used points of the template's matrix and TILE_PREBUILT[22][6]

A line of a census file: "ci interp sample_bytes fm wt probe" -- one per distinct instantiation of the fused kernel and process.  The table
holds, per launch configuration (row) and interpolator flag (column), a 4-bit mask over (fm + 2 * wt); probes are always prebuilt.
"""
import sys
from collections import defaultdict

used = defaultdict(set)
for path in sys.argv[1:]:
    for line in open(path):
        p = line.split()
        if len(p) != 6:
            continue
        ci, interp, sb, fm, wt, probe = map(int, p)
        if probe:
            continue
        used[(ci, interp)].add(fm + 2 * wt)
rows = []
n = 0
for ci in range(22):
    row = []
    for interp in range(6):
        m = 0
        for b in used.get((ci, interp), ()):
            m |= 1 << b
            n += 1
        row.append(m)
    rows.append(row)
print(f"// {n} instantiations used")
print("static constexpr unsigned char TILE_PREBUILT[22][6] = {")
for ci, row in enumerate(rows):
    print("    {" + ", ".join(f"0x{m:x}" for m in row) + "}," + f"   // cfg {ci}")
print("};")
