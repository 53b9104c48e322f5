import sys, ctypes as C
sys.path.insert(0,'.')
import alvaar_amd
from alvaar_amd.capi import lib, check
ctx = alvaar_amd.Context(0)
a, b = C.c_double(0), C.c_double(0)
lib.alva_microbench_peaks.argtypes=[C.c_void_p, C.c_void_p, C.c_void_p]
for _ in range(3):
    check(lib.alva_microbench_peaks(ctx.h, C.byref(a), C.byref(b)))
    print("FP64 MFMA %.1f TFLOP/s   int VALU %.1f Tops/s" % (a.value, b.value))
