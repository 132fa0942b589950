"""The C++ drop-in headers (include/SeqLib/*.h): compiled with g++ against libseqlib_amd.so and driven the way
a SeqLib user would (tests/cpp/seqlib_api_test.cpp).  CPU part mirrors the reference's own
tests/test_BamRecord.cpp:9-66 and the setter / accessor checks of seq_test/seq_test.cpp:798-883; the GPU part
checks per-read alignSequence and batch alignSequences against the committed golden records."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    import __graft_entry__ as g
    if not os.path.exists(os.path.join(ROOT, "seqlib_amd", "libseqlib_amd.so")):
        g.build()
    out = str(tmp_path_factory.mktemp("cpp") / "seqlib_api_test")
    lib = os.path.join(ROOT, "seqlib_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "seqlib_api_test.cpp"), "-o", out, "-L" + lib, "-lseqlib_amd",
                           "-Wl,-rpath," + lib])
    return out


def test_cpp_headers_cpu(exe, golden_dir, tmp_path):
    r = subprocess.run([exe, "cpu", os.path.join(golden_dir, "tiny.fa"), str(tmp_path / "rt")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "cpu checks OK" in r.stdout


@pytest.mark.gpu
def test_cpp_align_matches_golden(exe, golden_dir):
    n1, n2 = 40, 960
    r = subprocess.run([exe, "gpu", os.path.join(golden_dir, "tiny.fa"), os.path.join(golden_dir, "sim1_bcr.head3000.fq"), str(n1), str(n2)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = [l.split("\t") for l in r.stdout.strip().split("\n")]
    exp = [l.rstrip("\n").split("\t") for l in open(os.path.join(golden_dir, "sim1_head3000.records.tsv")) if int(l.split("\t")[0]) < n1 + n2]
    assert [g[:10] for g in got] == exp
    # record materialisation: forward-strand records carry the read itself; reverse ones the reference's A<->T-only reversal
    from oracle import orc
    _, seqs = orc.read_fastq(os.path.join(golden_dir, "sim1_bcr.head3000.fq"), n1 + n2)
    for g in got:
        s = seqs[int(g[0])]
        if int(g[2]) & 16:
            assert g[10] == s[::-1].translate(str.maketrans("AT", "TA"))
        else:
            assert g[10] == s
    assert "1 record(s), qname name" in r.stderr
