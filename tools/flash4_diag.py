"""Timing-only builds of the 4-wave flash kernel (make VARIANT=d VARIANT_FLAGS=-DOMX_F4_DIAG=1; OMX_LIB_VARIANT=d): what the FLUX shape costs
without the LDS-DMA, without the softmax VALU, without the LDS fragment reads, and with MFMAs alone (results are garbage by construction)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import omx_import
omx = omx_import.load_package()
T = omx.ops.Tensor
names = {0: "full", 2: "no DMA", 3: "no softmax VALU", 4: "no LDS reads", 5: "MFMA only", 6: "exp -> mov", 7: "no barrier", 8: "no row-sum adds"}
for name, (H, Hkv, S) in {"flux 24x4608": (24, 24, 4608), "8x2048": (8, 8, 2048)}.items():
    D = 128
    q = omx.ops.fill_uniform((1, H, S, D), 1, 1.0); k = omx.ops.fill_uniform((1, Hkv, S, D), 2, 1.0); v = omx.ops.fill_uniform((1, Hkv, S, D), 3, 1.0)
    out = T((1, H, S, D), "bf16")
    os.environ["OMX_ATTN_W4"] = "1"
    for rep in range(3):
        for var in (0, 2, 3, 4, 5, 6, 7, 8):
            os.environ["OMX_ATTN_W4_VAR"] = str(var)
            def run(n):
                for _ in range(n):
                    omx.check(omx.lib.omx_sdpa(out.ptr, q.ptr, k.ptr, v.ptr, 1, H, Hkv, S, S, D, Hkv * S * D, S * D, D ** -0.5, 0, None, 12, None))
                omx.ops.synchronize()
            run(3)
            t = time.perf_counter(); run(20); dt = (time.perf_counter() - t) / 20
            units = H * ((S + 255) // 256); rounds = -(-units // 256); nt = S // 64
            print(f"{name:14s} {names[var]:16s} {dt * 1e6:8.1f} us   {dt * 1e9 / (rounds * nt):7.1f} ns per tile step", flush=True)
