"""The C ABI from a plain-C host (no Python, no torch types): tests/c_abi/host_demo.c is compiled as C99 against
include/ctn_hip.h and linked with libctn_hip.so.  CPU: it compiles and links (the header is valid C and every symbol
it uses resolves).  GPU: it runs the reference's Encoder through the library and checks it against a scalar loop."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c_abi", "host_demo.c")
LIBDIR = os.path.join(ROOT, "conv-tasnet_amd")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")


def _build(out):
    import __graft_entry__ as g
    if not os.path.exists(os.path.join(LIBDIR, "libctn_hip.so")):
        g.build()
    cc = shutil.which("gcc") or shutil.which("cc")
    if cc is None:
        pytest.skip("no C compiler")
    cmd = [cc, "-std=c99", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROCM, "include"),
           SRC, "-o", out, "-L", LIBDIR, "-lctn_hip", "-L", os.path.join(ROCM, "lib"), "-lamdhip64", "-lm",
           "-Wl,-rpath," + LIBDIR, "-Wl,-rpath," + os.path.join(ROCM, "lib")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return out


def test_c_host_compiles_and_links(tmp_path):
    exe = _build(str(tmp_path / "host_demo"))
    assert os.path.getsize(exe) > 0


@pytest.mark.gpu
def test_c_host_runs_encoder(tmp_path):
    exe = _build(str(tmp_path / "host_demo"))
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.strip().endswith("OK"), r.stdout
