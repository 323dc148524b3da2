"""The C oracle under AddressSanitizer + UBSan (CPU only; GPU sanitizers are not available on the pool)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_selftest_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "selftest")
    subprocess.check_call(["gcc", "-O1", "-g", "-std=c11", "-Wall", "-Wextra", "-ffp-contract=off",
                           "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           os.path.join(ROOT, "oracle", "selftest.c"), os.path.join(ROOT, "oracle", "vfik_oracle.c"),
                           "-lm", "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "selftest OK" in r.stdout
