"""Cycle budget of the two-phase flash-attention kernel (csrc/attn_prefill.hip attn_prefill_pp_kernel, timeline build): per wave, shader
cycles spent in X work (QK^T + PV MFMAs and their LDS reads), Y work (softmax), and waiting at the even / odd phase barriers.
FLUX.2-klein joint-attention shape (24 heads x 128, 4608 tokens).  usage: python tools/attn_pp_trace.py [S] [H]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import omx_import
omx = omx_import.load_package()
T = omx.ops.Tensor
S = int(sys.argv[1]) if len(sys.argv) > 1 else 4608
H = int(sys.argv[2]) if len(sys.argv) > 2 else 24
D = 128
q = omx.ops.fill_uniform((1, H, S, D), 1, 1.0); k = omx.ops.fill_uniform((1, H, S, D), 2, 1.0); v = omx.ops.fill_uniform((1, H, S, D), 3, 1.0)
out = T((1, H, S, D), "bf16")
nblk = (S + 255) // 256 * H
tr = T((nblk * 8 * 8 * 2,), "u32")   # u64 words as u32 pairs
def run():
    omx.check(omx.lib.omx_sdpa(out.ptr, q.ptr, k.ptr, v.ptr, 1, H, H, S, S, D, H * S * D, S * D, D ** -0.5, 0, None, 12, None))
    omx.ops.synchronize()
os.environ["OMX_ATTN_PP"] = "1"
run()
os.environ["OMX_ATTN_PP_TRACE"] = hex(tr.ptr)
run(); run()
t = tr.numpy().view(np.uint64).reshape(nblk, 8, 8).astype(np.float64)
nt = t[0, 0, 5]
print(f"{nblk} blocks, {int(nt)} key tiles each; cycles PER TILE STEP (two phases), mean over blocks")
print("waves    X work   Y work  wait@even  wait@odd   total")
for name, sl in (("0-3 (early)", slice(0, 4)), ("4-7 (late)", slice(4, 8))):
    m = t[:, sl, :5].mean(axis=(0, 1)) / (nt + 1)
    print(f"{name:12s} {m[0]:7.0f} {m[1]:8.0f} {m[2]:9.0f} {m[3]:9.0f} {m[4]:8.0f}")
first = t[:256, :, 4].mean() ; last = t[256:, :, 4].mean() if nblk > 256 else float('nan')
print(f"block total cycles: first 256 blocks {first:.0f}, the rest {last:.0f}")
