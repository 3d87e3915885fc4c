import ctypes as C, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(ROOT, "proxmin_amd", "libpmx_floor.so"))
lib.pmxf_last_error.restype = C.c_char_p
names = {0: "one accumulator, back to back", 1: "one accumulator, 2 LDS reads per 3 MFMAs", 2: "two accumulators alternating, same reads", 3: "four accumulators, same reads"}
for waves in (1, 2):
    for mode in (0, 1, 2, 3):
        v = C.c_double()
        rc = lib.pmxf_chain(0, mode, waves, 20, C.byref(v))
        print("waves/SIMD %d | %-45s | %.2f ns per MFMA of a wave%s" % (waves, names[mode], v.value, "" if rc == 0 else "  ERR " + lib.pmxf_last_error().decode()))
