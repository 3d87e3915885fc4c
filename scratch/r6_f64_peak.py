import ctypes as C, os, sys
lib = C.CDLL(os.path.join(os.getcwd(), "proxmin_amd", "libpmx_floor.so"))
lib.pmxf_last_error.restype = C.c_char_p
v = C.c_double()
for waves in (1, 2):
    for rnd in (1, 0):
        rc = lib.pmxf_mfma_f64(0, rnd, waves, 3, C.byref(v))
        print("fp64 MFMA 16x16x4: %d wave(s)/SIMD, %s data: %.1f TFLOP/s (rc %d)" % (waves, "random" if rnd else "zero", v.value, rc))
