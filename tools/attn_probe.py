import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import omx_import
omx = omx_import.load_package()
T = omx.ops.Tensor
for Tk in (64, 256, 2304):
    q = omx.ops.fill_uniform((1, 32, 1, 128), 1, 1.0)
    k = omx.ops.fill_uniform((1, 8, Tk, 128), 2, 1.0)
    v = omx.ops.fill_uniform((1, 8, Tk, 128), 3, 1.0)
    for _ in range(50):
        o = omx.ops.scaled_dot_product_attention(q, k, v, 0.088)
    omx.ops.synchronize()
