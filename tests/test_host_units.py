"""Host builds of the algorithm headers the device code is made of (same source, g++):
the libstdc++-exact selection and the glibc-exact cosf/sinf."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("native") / "host_units")
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-std=c++17", "-pthread", "-o", out,
                           os.path.join(HERE, "native", "host_units.cpp")])
    return out


def test_selection_matches_std_nth_element(exe):
    r = subprocess.run([exe, "select"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert " 0 mismatching" in r.stdout


def test_trig_port_matches_host_libm(exe):
    # every 13th float of [0, 6.2832] (the exhaustive sweep, stride 1, takes ~6 s on 8 cores and also passes)
    r = subprocess.run([exe, "trig", "13"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "fma=0 stride=13: sin mismatches 0, cos mismatches 0" in r.stdout
