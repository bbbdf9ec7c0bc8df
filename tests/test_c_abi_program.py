"""The C ABI from C: tests/c/abi_demo.c is compiled as strict C99 against include/xmhw_amd.h and
linked to libxmhw_amd.so (CPU: proves the header is plain C and every symbol it uses resolves);
on a GPU box the program runs (one-shot host entry point == resident-data entry points)."""
import os
import subprocess

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
SRC = os.path.join(ROOT, "tests", "c", "abi_demo.c")
LIBDIR = os.path.join(ROOT, "xmhw_amd")


def _build(outdir):
    exe = os.path.join(str(outdir), "abi_demo")
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), SRC,
           "-L", LIBDIR, "-lxmhw_amd", "-lm", "-Wl,-rpath," + LIBDIR, "-o", exe]
    subprocess.check_call(cmd)
    return exe


def test_header_is_plain_c_and_links(tmp_path):
    if not os.path.exists(os.path.join(LIBDIR, "libxmhw_amd.so")):
        pytest.skip("libxmhw_amd.so not built")
    exe = _build(tmp_path)
    assert os.path.exists(exe)


@pytest.mark.gpu
def test_c_program_runs(tmp_path):
    exe = _build(tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "arch gfx950" in out.stdout and out.stdout.strip().endswith("ok")
    assert "refused:" in out.stdout
