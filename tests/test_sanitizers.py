"""AddressSanitizer + UndefinedBehaviorSanitizer runs of the CPU code (the reference builds its own tests with
`-fsanitize=address,undefined`, /root/reference/test_build.sh:1): the oracle end to end on fixture reads -- results equal to the
regular build's -- and the product's host-side index code (file formats, .alt parsing).  GPU sanitizers are not available on the
pool; device code is covered by the parity tests."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-g", "-O1"]
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")


def test_oracle_under_asan_ubsan(golden_dir, tmp_path, orc):
    exe = str(tmp_path / "san_oracle")
    o = os.path.join(ROOT, "oracle")
    objs = []
    for src, cc, std in (("orc_index.c", "gcc", "-std=gnu11"), ("orc_mem.c", "gcc", "-std=gnu11"), ("orc_glue.cpp", "g++", "-std=c++17")):
        obj = str(tmp_path / (src + ".o"))
        subprocess.check_call([cc, std] + SAN + ["-ffp-contract=off", "-I" + o, "-c", os.path.join(o, src), "-o", obj])
        objs.append(obj)
    obj = str(tmp_path / "main.o")
    subprocess.check_call(["gcc", "-std=gnu11"] + SAN + ["-I" + o, "-c", os.path.join(ROOT, "tests", "cpp", "san_oracle_test.c"), "-o", obj])
    subprocess.check_call(["g++"] + SAN + objs + [obj, "-o", exe, "-lm", "-lpthread"])
    n = 400
    fq = os.path.join(golden_dir, "sim2_bcr.head3000.fq")
    r = subprocess.run([exe, os.path.join(golden_dir, "tiny.fa"), fq, str(n), str(tmp_path / "rt")], capture_output=True, text=True, env=ENV, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr
    # same records as the regular (-O2) build
    _, seqs = orc.read_fastq(fq, n)
    idx = orc.Index.load(os.path.join(golden_dir, "tiny.fa"))
    recs = sum(len(orc.align_sequence(orc.default_opt(), idx, s, hardclip=bool(i & 1), ordinal=i)) for i, s in enumerate(seqs))
    assert int(r.stdout.split()[0]) == recs


def test_index_host_code_under_asan_ubsan(golden_dir, tmp_path):
    exe = str(tmp_path / "san_index")
    c = os.path.join(ROOT, "seqlib_amd", "csrc")
    subprocess.check_call(["g++", "-std=c++17"] + SAN + ["-I" + os.path.join(ROOT, "include"), "-I" + c, os.path.join(c, "slx_index.cpp"),
                           os.path.join(ROOT, "tests", "cpp", "san_index_test.cpp"), "-o", exe])
    r = subprocess.run([exe, os.path.join(golden_dir, "tiny.fa"), str(tmp_path / "w")], capture_output=True, text=True, env=ENV, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr
    assert r.stdout.strip() == "nseq=4 l_pac=354751 alt=1"
