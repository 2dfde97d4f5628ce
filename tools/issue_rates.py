#!/usr/bin/env python3
"""csrc/probe.hip on this device: the saturated VALU issue rates bench.py's roofline uses (mixes 0-2) and the three pair-loop forms of tools/coef_window.py with their
LDS gathers (mixes 3-5: ns per TRANSMIT PAIR of four traces and wave slot, four waves per SIMD on every CU).  Run through gpurun; profiles/r05/coef_window.txt."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qups_amd import _lib
L = _lib.lib()
f = L.qdas_debug_issue_rate
f.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
names = ["broadcast v_pk_fma_f32", "v_fma_f32", "pair loop's VALU mix, no LDS (37 per transmit pair)",
         "taps + weights, folded + mirror loop: 37 VALU + 16 ds_read_b64 per transmit pair",
         "coefficient windows, cubic (D = 3): 23 VALU + 8 ds_read_b128 per transmit pair",
         "coefficient windows, degree 7 (lanczos3): 39 VALU + 16 ds_read_b128 per transmit pair"]
res = []
for rep in range(3):
    row = []
    for mix in range(6):
        v, ms = C.c_double(), C.c_double()
        rc = f(-1, mix, C.byref(v), C.byref(ms))
        assert rc == 0, (mix, rc)
        row.append((v.value, ms.value))
    res.append(row)
print("mix  ns per unit and SIMD (3 runs)      launch ms   unit")
for mix in range(6):
    vals = [res[r][mix][0] for r in range(3)]
    unit = "wave64 instruction" if mix < 3 else "transmit pair (4 traces x 64 pixels)"
    print(f" {mix}   " + "  ".join(f"{v:8.3f}" for v in vals) + f"   {res[0][mix][1]:8.2f}   {unit}: {names[mix]}")
t3, t4, t5 = (min(res[r][m][0] for r in range(3)) for m in (3, 4, 5))
print(f"coefficient form / tap form, time per transmit pair: cubic {t4 / t3:.3f}, degree 7 {t5 / t3:.3f}  (loop only: no conversion of staged samples, no extra staging)")
