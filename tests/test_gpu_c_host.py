"""The C-ABI from a plain C99 host (tests/c_host/c_host.c): no Python, no torch in the process that drives
the library.  One cycle through vfik_step_host and a 50-cycle vfik_rollout on device pointers, both
checked against the oracle inside the C program."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_c_host_program_runs_and_matches_the_oracle():
    import __graft_entry__ as g
    g.build()
    d = os.path.join(ROOT, "tests", "c_host")
    subprocess.check_call(["make", "-s", "-C", d, "c_host"])
    r = subprocess.run([os.path.join(d, "c_host")], capture_output=True, text=True, timeout=120)
    print(r.stdout, r.stderr)
    assert r.returncode == 0, (r.stdout, r.stderr)
    assert "c_host OK" in r.stdout


def test_headers_are_valid_c99_and_the_host_links():
    """include/*.h compile as strict C99 (-pedantic -Werror) and every entry point the C host uses resolves."""
    import __graft_entry__ as g
    g.build()
    d = os.path.join(ROOT, "tests", "c_host")
    subprocess.check_call(["make", "-s", "-C", d, "clean"])
    subprocess.check_call(["make", "-s", "-C", d, "c_host"])
    assert os.path.exists(os.path.join(d, "c_host"))
