import ctypes as C, torch
from pointcontrast_amd._lib import lib
a, b, c = C.c_int(), C.c_int(), C.c_int()
print("rc", lib.pcmi_debug_conv_occupancy(C.byref(a), C.byref(b), C.byref(c)), "conv16<3>", a.value, "conv16<4>", b.value, "mfma<3,sk>", c.value)
