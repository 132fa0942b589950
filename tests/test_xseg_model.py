"""The algorithm behind the contigs' segmented extensions (seqlib_amd/csrc/dev_ext_seg.h) in scalar C (tests/second/xseg_model.c): segments started early
from a neutral state, verified against the true window at their first row, computed again when the verification fails -- equal to the CPU checker's
plain ksw_extend2 on every one of a few thousand seeded cases (substitutions, small and large indels, tandem repeats, diverged tails that end by z-drop,
Ns, narrow bands, other gap costs, small and large h0), with both outcomes of the verification exercised."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def exe(orc, tmp_path_factory):
    out = str(tmp_path_factory.mktemp("xseg") / "xseg_model")
    lib = os.path.join(ROOT, "oracle")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-o", out, os.path.join(ROOT, "tests", "second", "xseg_model.c"), "-L" + lib, "-lorc", "-Wl,-rpath," + lib])
    return out


@pytest.mark.parametrize("args", [("500", "1"), ("500", "2", "512", "160", "16"), ("200", "3", "4096", "512", "32"), ("300", "4", "256", "128", "4"), ("300", "5", "1024", "64", "8")])
def test_segmented_extension_equals_scalar(exe, args):
    r = subprocess.run([exe] + list(args), capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["bad"] == 0 and d["cases"] == int(args[0])
    if d["SEG"] <= 1024:
        assert d["spec_ok"] > 100 and d["fallback_state"] + d["fallback_event"] > 10, d          # both outcomes of the verification were taken
