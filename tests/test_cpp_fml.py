"""The C++ mirrors of SeqLib::FermiAssembler and SeqLib::BFC (include/SeqLib/FermiAssembler.h, BFC.h) compiled with g++ against
libseqlib_amd.so and driven as the reference's own callers drive them (tests/cpp/fml_api_test.cpp); the GPU part is compared with the
CPU checker record for record."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    import __graft_entry__ as g
    if not os.path.exists(os.path.join(ROOT, "seqlib_amd", "libseqlib_amd.so")):
        g.build()
    out = str(tmp_path_factory.mktemp("cpp") / "fml_api_test")
    lib = os.path.join(ROOT, "seqlib_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "fml_api_test.cpp"),
                           "-o", out, "-L" + lib, "-lseqlib_amd", "-Wl,-rpath," + lib, "-lz", "-lpthread"])
    return out


def test_cpp_fml_headers_cpu(exe):
    r = subprocess.run([exe, "cpu"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "cpu checks OK" in r.stdout
    import torch
    if not torch.cuda.is_available():
        assert "no-GPU call throws" in r.stdout          # no CPU fallback behind the classes either


@pytest.mark.gpu
def test_cpp_fml_pipeline_matches_oracle(exe, golden_dir):
    from oracle import orc_fml as F
    n = 3000
    fq = os.path.join(golden_dir, "sim1_bcr.head3000.fq")
    r = subprocess.run([exe, "gpu", fq, str(n)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = {}
    for line in r.stdout.split("\n"):
        if line:
            k, _, v = line.partition("\t")
            out.setdefault(k, []).append(v)
    L = open(fq).read().split("\n")
    names = [L[i][1:].split()[0] for i in range(0, 4 * n, 4)]
    seqs = [L[i + 1].encode() for i in range(0, 4 * n, 4)]
    quals = [L[i + 3].encode() for i in range(0, 4 * n, 4)]
    # 1. CorrectReads, then PerformAssembly on the corrected reads (fml_assemble corrects them once more, as the reference's pipeline does)
    R = F.Reads(seqs, quals)
    o = F.default_opt()
    F.opt_adjust(o, R)       # CorrectReads with ec_k = 0 means fml_opt_adjust's k here (the reference hands fermi-lite k = 0: a shift by -1)
    F.correct(o, R)
    cs, cq = R.get()
    assert out["COR"] == ["%s\t%s" % (nm, s.decode()) for nm, s in zip(names, cs)]
    exp = F.assemble(F.default_opt(), F.Reads(cs, cq))
    assert out.get("CTG", []) == [u["seq"].decode() for u in exp] and len(exp) > 5
    assert int(out["GFA"][0]) > sum(u["len"] for u in exp)
    # 2. BFC Train / ErrorCorrect, DirectAssemble
    R = F.Reads(seqs, quals)
    o = F.default_opt(); F.opt_adjust(o, R)
    c = F.Count(R, o.ec_k)
    kcov, _ = c.error_correct(F.default_opt(), R)
    k, kc = out["BFC"][0].split("\t")
    assert int(k) == o.ec_k and abs(float(kc) - kcov) < 1e-3 * kcov
    es, _ = R.get()
    d = F.default_opt()
    exp = F.direct_assemble(d, kcov, F.Reads([s.upper() for s in es]))
    assert out.get("DIR", []) == [u["seq"].decode() for u in exp]
    # 3. two windows in one call
    for w, (a, b) in enumerate(((0, n // 2), (n // 2, n))):
        exp = F.assemble(F.default_opt(), F.Reads(seqs[a:b], quals[a:b]))
        assert out.get("WIN%d" % w, []) == [u["seq"].decode() for u in exp]
    # 4. CorrectAndFilterReads
    R = F.Reads(seqs, quals)
    o = F.default_opt(); F.opt_adjust(o, R)
    F.fltuniq(o, R)
    fs, _ = R.get()
    assert out["FLT"] == [s.decode() for s in fs]
    # 5. two BFC objects, each correcting against its own table, with a FermiAssembler call in between (the table is the object's)
    half = n // 2
    got_k = out["BF2"][0].split("\t")
    for tag, (a, b), k, gk, gkcov in (("BFA", (0, half), 17, got_k[0], got_k[1]), ("BFB", (half, n), 21, got_k[2], got_k[3])):
        R = F.Reads(seqs[a:b], quals[a:b])
        c = F.Count(R, k)
        kcov, _ = c.error_correct(F.default_opt(), R)
        es, _ = R.get()
        assert int(gk) == k and abs(float(gkcov) - kcov) < 1e-3 * kcov
        assert out[tag] == [s.decode().upper() for s in es], tag
    # 6. every third read without a quality string: those reads count as all-high-quality, the others keep their own qualities
    R = F.Reads(seqs, quals)
    for i in range(1, n, 3):
        R.drop_qual(i)
    o = F.default_opt(); F.opt_adjust(o, R)
    F.correct(o, R)
    ms, _ = R.get()
    assert out["MIX"] == [s.decode() for s in ms]


@pytest.mark.gpu
def test_cpp_assemble_then_realign_like_seqtools(exe, golden_dir):
    """the whole of BASELINE config 5 through the reference's own classes (src/seqtools/seqtools.cpp:106-212): reads into one FermiAssembler,
    CorrectReads, PerformAssembly, every contig through BWAAligner::alignSequence -- contigs and records equal to what the two CPU checkers
    give for the same calls (the i-th alignSequence call of the process takes lrand48 draw i)"""
    from oracle import orc, orc_fml as F
    n = 3000
    fq = os.path.join(golden_dir, "sim2_bcr.head3000.fq")
    prefix = os.path.join(golden_dir, "tiny.fa")
    r = subprocess.run([exe, "pipeline", prefix, fq, str(n)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    L = open(fq).read().split("\n")
    seqs = [L[i + 1].encode() for i in range(0, 4 * n, 4)]
    quals = [L[i + 3].encode() for i in range(0, 4 * n, 4)]
    R = F.Reads(seqs, quals)
    o = F.default_opt()
    F.opt_adjust(o, R)
    F.correct(o, R)
    cs, cq = R.get()
    contigs = [u["seq"].decode() for u in F.assemble(F.default_opt(), F.Reads(cs, cq))]
    oidx = orc.Index.load(prefix)
    opt = orc.default_opt()
    exp = []
    for i, c in enumerate(contigs):
        recs = orc.align_sequence(opt, oidx, c, name="contig%d" % i, ordinal=i)
        exp.append("CTG\t%d\t%d\t%d" % (i, len(c), len(recs)))
        exp += ["REC\t%d\t%d\t%d\t%d\t%d\t%s" % (i, x["rid"], x["pos"], x["flag"], x["mapq"], orc.cigar_str(x["cigar"])) for x in recs]
    got = [ln for ln in r.stdout.split("\n") if ln]
    assert got == exp
    assert len(contigs) > 5 and max(len(c) for c in contigs) > 1000 and sum(ln.startswith("REC") for ln in got) >= len(contigs) // 2
