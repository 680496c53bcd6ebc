#!/usr/bin/env python3
"""Prints VGPR / scratch / occupancy of the dense kernels from `make -C bsmr-sddmm_amd asm`."""
import re
import sys
from pathlib import Path

txt = (Path(__file__).resolve().parent.parent / "bsmr-sddmm_amd/build/resource_usage.txt").read_text()
pat = sys.argv[1] if len(sys.argv) > 1 else "denseGroupsILi"
for b in re.split(r"remark: Function Name: ", txt)[1:]:
    name = b.split(" ")[0]
    if pat not in name:
        continue
    get = lambda k: re.search(re.escape(k) + r"[^:]*: (\d+)", b).group(1)
    print(name[:70], "VGPR", get("VGPRs"), "AGPR", get("AGPRs"), "SGPR", get("TotalSGPRs"), "scratch",
          get("ScratchSize"), "occ", get("Occupancy"))
