"""Per-block timeline of the decode GEMVs (GemvArgs::trace): a chain of back-to-back launches of one shape on distinct weight
buffers; reports, relative to the first block start of each launch: block starts, activation staged, first batch reduced, block
end -- and the boundary between consecutive launches (last end -> next first start).
usage: python tools/gemv_trace.py [rpw overrides as name=rpw ...]"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import omx_import
omx = omx_import.load_package()
lib = omx.lib   # needs OMX_LIB_VARIANT=trace (make VARIANT=trace VARIANT_FLAGS=-DOMX_GEMV_TRACE)
lib.omx_bench_gemv_trace.restype = ctypes.c_int
lib.omx_bench_gemv_trace.argtypes = [ctypes.c_int] * 6 + [ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_int)]
PRO = {"none": 0, "rms": 1}
EPI = {"store": 0, "resid": 1, "swiglu": 2, "argmax": 3, "f32": 4}
shapes = [("qkv", 6144, 4096, "rms", "store"), ("o_proj", 4096, 4096, "none", "resid"), ("gate_up", 12288, 4096, "rms", "swiglu"),
          ("down", 4096, 12288, "none", "resid")]
over = dict(a.split("=") for a in sys.argv[1:])
chain = 6
for name, N, K, pro, epi in shapes:
    for rpw in [int(v) for v in over.get(name, "0").split(",")]:
        buf = np.zeros(chain * 8192 * 4, np.uint64)
        nb = ctypes.c_int()
        omx.check(lib.omx_bench_gemv_trace(N, K, PRO[pro], EPI[epi], rpw, chain, buf.ctypes.data, buf.size, ctypes.byref(nb)))
        nb = nb.value
        t = buf[:chain * nb * 4].reshape(chain, nb, 4).astype(np.int64) / 100.0
        mb = N * K * 2 * (2 if epi == "swiglu" else 1) / 1e6
        spans, gaps, rel = [], [], []
        for i in range(1, chain):
            t0 = t[i, :, 0].min()
            spans.append(t[i, :, 3].max() - t0)
            gaps.append(t0 - t[i - 1, :, 3].max())
            rel.append(t[i] - t0)
        rel = np.stack(rel)
        q = lambda a: f"med {np.median(a):5.2f} p90 {np.percentile(a, 90):5.2f} max {np.median(a.max(axis=1)):5.2f}"
        print(f"{name:8s} rpw {rpw} blocks {nb} {mb:6.1f} MB  span {np.median(spans):6.2f} us ({mb / np.median(spans) / 1e3:5.2f} TB/s)  "
              f"gap to next launch {np.median(gaps):5.2f} us  -> period {np.median(spans) + np.median(gaps):6.2f} us")
        print(f"           start {q(rel[:, :, 0])} | x staged {q(rel[:, :, 1])} | first batch {q(rel[:, :, 2])} | end {q(rel[:, :, 3])}"
              f"  end min {np.median(rel[:, :, 3].min(axis=1)):5.2f}")
        # workgroup b runs on XCD b % 8 (tools/xcc_probe.py): is the tail an XCD effect?
        xe = [np.median(rel[:, x::8, 3].max(axis=1)) for x in range(8)]
        xm = [np.median(rel[:, x::8, 3]) for x in range(8)]
        print("           last end per XCD " + " ".join(f"{v:5.2f}" for v in xe) + " | median end per XCD " + " ".join(f"{v:5.2f}" for v in xm))
