"""The oracle against every known-answer test the reference holds for the render path.

oracle/kat.cpp restates the reference's `test` blocks (inputs, expected values and
tolerances from the .zig files); this module builds and runs it and reports one pytest
case per reference test group."""
import subprocess

import pytest

import oracle_lib as O


def _run_kat():
    O.lib()  # builds if needed
    p = subprocess.run([O.KAT_PATH], capture_output=True, text=True)
    return p.returncode, p.stdout.splitlines()


_RC, _LINES = _run_kat()
_CASES = [ln for ln in _LINES if ln.startswith(("ok ", "FAIL "))]


def test_kat_binary_passes():
    assert _RC == 0, "\n".join(ln for ln in _LINES if ln.startswith("FAIL"))
    assert len(_CASES) >= 50
    assert _LINES[-1].startswith("KAT PASSED")


@pytest.mark.parametrize("line", _CASES, ids=[c.split(" ", 1)[1][:70] for c in _CASES])
def test_kat_case(line):
    assert line.startswith("ok "), line
