"""The generated kernel bodies in the tree are what their generators emit (CPU): `ominix-mlx_amd/csrc/*.inc` is committed so that the library builds
without running Python, and a generator edited without regenerating -- or an .inc edited by hand -- would otherwise go unnoticed until a GPU run."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ominix-mlx_amd", "csrc")


@pytest.mark.parametrize("script,files", [
    ("gen_gemm5_asm.py", ["gemm5_body.inc"]),
    ("gen_flash4_asm.py", ["attn_flash4_body.inc", "attn_flash4_body_thr0.inc", "attn_flash4_clobbers.inc"]),
    ("gen_gemm4_asm.py", ["gemm4_body.inc"]),
])
def test_committed_body_is_the_generators_output(tmp_path, script, files):
    env = {k: v for k, v in os.environ.items() if not k.startswith("G5_")}      # (the generators' experiment switches)
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", script), "--out", str(tmp_path)], check=True, env=env, stdout=subprocess.DEVNULL)
    for name in files:
        assert open(os.path.join(tmp_path, name)).read() == open(os.path.join(CSRC, name)).read(), f"{name}: run tools/{script} and rebuild"
