"""Cost of one device-wide barrier (csrc/gridsync.hpp) on the GPU: tools/barrier_probe.py"""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import omx_import
omx = omx_import.load_package()
lib = omx.lib
lib.omx_bench_grid_barrier.restype = ctypes.c_int
lib.omx_bench_grid_barrier.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int)]
for var, name in ((0, "flags + coherent accessors (gridsync.hpp)"), (1, "atomic counter + agent fences"),
                  (2, "XCD-local groups through their own L2 (sc0)")):
    for nb in (256, 512):
        us = ctypes.c_float(); bad = ctypes.c_int()
        omx.check(lib.omx_bench_grid_barrier(nb, 500, var, ctypes.byref(us), ctypes.byref(bad)))
        print(f"{name:45s} blocks {nb:4d}: {us.value:7.3f} us per exchange+barrier  failed={bad.value}", flush=True)
