#!/usr/bin/env python3
"""Register / spill table of every das_tile_kernel instantiation: tools/kernel_regs.py <device code object>
(hipcc -DQDAS_UNITY --cuda-device-only -c das_tile.hip -o tile.co, then clang-offload-bundler --unbundle; reads the AMDGPU metadata notes)."""
import re, subprocess, sys
t = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", sys.argv[1]], capture_output=True, text=True).stdout
for b in t.split("- .agpr_count")[1:]:
    g = lambda k: (re.search(r"\." + k + r":\s+(\d+)", b) or [None, "?"])[1]
    name = re.search(r"\.name:\s+(\S+)", b)
    if not name or "das_tile_kernel" not in name.group(1):
        continue
    m = re.search(r"ILi(\d)E(\w+?)Lb(\d)ELb(\d)ELb(\d)ELi(\d+)ELi(\d+)ELi(\d+)ELi\d+ELi\d+ELi\d+ELb(\d)E", name.group(1))
    if not m:
        continue
    interp, st, fm, wt, sym, waves, mb, w, probe = m.groups()
    if probe == "1":
        continue
    print(f"interp {interp} {'f16' if st == 'j' else 'f32'} fmod {fm} wtab {wt} sym {sym} mb {mb} W {w}: vgpr {g('vgpr_count')} "
          f"vgpr_spill {g('vgpr_spill_count')} scratch {g('private_segment_fixed_size')} sgpr {g('sgpr_count')} sgpr_spill {g('sgpr_spill_count')}")
