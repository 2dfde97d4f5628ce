#!/usr/bin/env python3
"""Compile the plan-specialised (hiprtc) kernel of a given shape -- no GPU needed -- and disassemble it.

usage: jit_dump.py <interp 0|1|2|3|5> <dtype 1=f32|2=f16> <sym 0|1> <fmod 0|1> <N> <M> <T> ["has_apix=1,apix_real=1,kindB=2,tzl=6,wzl=3,I1=512,..."]
       (environment: QDAS_JIT_MB, QDAS_JIT_NARROW, QDAS_JIT_DEFINES as for plans)
Writes /tmp/qdas_jit_dump/<hash>.hsaco and <hash>.s and prints the register report of tools/kernel_regs.py."""
import ctypes as C, glob, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = "/tmp/qdas_jit_dump"
shutil.rmtree(d, ignore_errors=True)
os.environ["QDAS_CACHE_DIR"] = d
interp, dt, sym, fm, N, M, T = [int(v) for v in sys.argv[1:8]]
if len(sys.argv) > 8:
    os.environ["QDAS_JIT_DEBUG_SPEC"] = sys.argv[8]
L = C.CDLL(os.path.join(ROOT, "qups_amd", "libqdas.so"))
f = L.qdas_debug_jit_compile
f.argtypes = [C.c_int] * 4 + [C.c_ulonglong] * 3 + [C.c_char_p, C.c_size_t, C.POINTER(C.c_ulonglong)]
msg = C.create_string_buffer(4000)
n = C.c_ulonglong()
rc = f(interp, dt, sym, fm, N, M, T, msg, 4000, C.byref(n))
print("rc", rc, msg.value.decode()[:1000], n.value, "bytes")
if rc == 0:
    hs = glob.glob(d + "/*.hsaco")[0]
    subprocess.run(f"/opt/rocm/lib/llvm/bin/llvm-objdump -d {hs} > {hs[:-6]}.s", shell=True)
    print(subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_regs.py"), d], capture_output=True, text=True).stdout[-500:])
    print("disassembly:", hs[:-6] + ".s")
