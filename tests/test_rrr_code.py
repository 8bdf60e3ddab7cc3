"""The block code of the rrr-63 index variant (csrc/rrr_code.hpp) on the CPU: the same header the HIP kernels compile."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rrr_block_code_roundtrip(tmp_path):
    exe = str(tmp_path / "rrr_code_check")
    subprocess.check_call(["g++", "-O2", "-std=c++14", "-I", os.path.join(ROOT, "vlg_matching_amd", "csrc"), "-o", exe,
                           os.path.join(ROOT, "tests", "rrr_code_check.cpp")])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "class 3: 39711 blocks numbered 0..39710" in out.stdout and "class 60: 39711" in out.stdout
    assert out.stdout.strip().splitlines()[-1].startswith("ok ")
