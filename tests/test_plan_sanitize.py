"""The host-side plan builder (tracks, step tables, chunks: xmhw_amd/csrc/plan.cpp) compiled with
AddressSanitizer + UndefinedBehaviorSanitizer and driven over calendars, tstep axes and pathological
label sequences (sanitizers run on the CPU build only; the GPU pool does not offer them)."""
import os
import subprocess

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_plan_builder_under_asan_ubsan(tmp_path):
    exe = os.path.join(str(tmp_path), "plan_sanitize")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           os.path.join(ROOT, "tests", "c", "plan_sanitize.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "plan sanitize ok" in out.stdout
