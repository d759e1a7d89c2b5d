"""CPU-only: libomx_hip.so loads without a GPU and exports every symbol include/*.h declares;
the Python binding tables cover the same set (no compute calls here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b((?:omx|mlx)_[a-z0-9_]+)\s*\(", src)
    return sorted({n for n in names if not n.endswith("_func")})


def test_library_loads_without_gpu_and_reports_no_device(omx):
    assert omx.version().startswith("omx-hip")
    assert omx.device_count() >= 0


def test_every_declared_symbol_is_exported(omx):
    lib = ctypes.CDLL(omx.LIB_PATH)
    missing = [n for h in ("omx.h", "omx_mlx_c.h") for n in declared(h) if not hasattr(lib, n)]
    assert not missing, f"declared in include/ but not exported: {missing}"


def test_binding_tables_cover_the_headers(omx):
    from ominix_mlx_amd import audio, comm, engine, ep, klein, mlx_c, moe, paraformer, vae
    bound = set(omx.SIGNATURES) | set(engine.ENGINE_SIGNATURES) | set(mlx_c.SIGNATURES) | set(audio.AUDIO_SIGNATURES) | set(moe.MOE_SIGNATURES) | set(klein.KLEIN_SIGNATURES) | set(paraformer.PARAFORMER_SIGNATURES) | set(ep.EP_SIGNATURES) | set(comm.LOOPBACK_SIGNATURES) | set(vae.VAE_SIGNATURES)
    want = set(declared("omx.h")) | set(declared("omx_mlx_c.h"))
    assert want - bound == set(), f"no ctypes signature for: {sorted(want - bound)}"


def test_product_has_no_cpu_fallback(omx):
    """Compute entry points must fail loudly without a device (never route to the oracle)."""
    import pytest
    if omx.device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(omx.OmxError):
        omx.ops.Tensor((4,), "bf16")
    for mod in sorted(f for f in os.listdir(os.path.join(ROOT, "ominix-mlx_amd")) if f.endswith(".py")):
        text = open(os.path.join(ROOT, "ominix-mlx_amd", mod)).read()
        assert "import oracle" not in text and "from oracle" not in text
