"""CPU tests of the drop-in boundary: the C-ABI library loads, exports every symbol include/seqlib_amd.h
declares, its host-side pieces (options, index files, lrand48 helpers) agree with the oracle / the reference's
fixtures, and it FAILS LOUDLY when no GPU is present (no CPU fallback).  No compute call needs a GPU here."""
import ctypes as C
import filecmp
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ffi():
    import __graft_entry__ as g
    if not os.path.exists(os.path.join(ROOT, "seqlib_amd", "libseqlib_amd.so")):
        g.build()
    from seqlib_amd import _ffi
    _ffi.lib()
    return _ffi


def test_exports_match_header(ffi):
    hdr = open(os.path.join(ROOT, "include", "seqlib_amd.h")).read()
    body = hdr[hdr.index("extern \"C\""):]
    declared = set(re.findall(r"\b(slx_[a-z0-9_]+)\s*\(", body))
    assert declared == set(ffi.EXPORTS), declared ^ set(ffi.EXPORTS)
    L = ffi.lib()
    for name in declared:
        assert hasattr(L, name), name
    # every entry point cites the reference interface it replaces
    for name in ("slx_opt_init", "slx_index_build", "slx_index_load", "slx_index_write", "slx_aligner_create", "slx_align_batch"):
        assert name in hdr[:hdr.index("#ifndef")]
    assert "src/BWAAligner.cpp:89-146" in hdr and "src/BWAIndex.cpp:83-180" in hdr


def test_fml_exports_match_header(ffi):
    """include/seqlib_amd_fml.h (the FermiAssembler / BFC path): every declared symbol is exported and bound, every entry cites what it replaces"""
    from seqlib_amd import fml
    hdr = open(os.path.join(ROOT, "include", "seqlib_amd_fml.h")).read()
    body = hdr[hdr.index("extern \"C\""):]
    declared = set(re.findall(r"\b(slx_fml_[a-z0-9_]+)\s*\(", body))
    assert declared == set(fml.EXPORTS), declared ^ set(fml.EXPORTS)
    L = fml.lib()
    for name in declared:
        assert hasattr(L, name), name
    head = hdr[:hdr.index("#ifndef")]
    for name in ("slx_fml_opt_init", "slx_fml_correct", "slx_fml_count", "slx_fml_error_correct", "slx_fml_assemble", "slx_fml_direct_assemble"):
        assert name in head
    assert "src/FermiAssembler.cpp:133-138" in hdr and "src/BFC.cpp:262-270" in hdr and "src/FermiAssembler.cpp:26-44" in hdr


def test_fml_options_and_no_gpu(ffi):
    from oracle import orc_fml
    from seqlib_amd import fml
    o, e = fml.default_opt(), orc_fml.default_opt()
    for name, _ in fml.FmlOpt._fields_:
        if name != "mag_opt":
            assert getattr(o, name) == getattr(e, name), name
    for name, _ in fml.MagOpt._fields_:
        assert getattr(o.mag_opt, name) == getattr(e.mag_opt, name), name
    import numpy as np
    for n, L in ((8000, 150), (100000, 150), (2000, 100), (5, 100)):
        o, e = fml.default_opt(), orc_fml.default_opt()
        lens = np.full(n, L, dtype=np.int32)
        fml.lib().slx_fml_opt_adjust(C.byref(o), n, lens.ctypes.data)
        R = orc_fml.Reads([b"A" * L] * n)
        orc_fml.opt_adjust(e, R)
        assert (o.ec_k, o.mag_opt.min_elen) == (e.ec_k, e.mag_opt.min_elen)
    import torch
    if not torch.cuda.is_available():
        with pytest.raises(ffi.SlxError) as ex:
            fml.Context()
        assert ex.value.code == ffi.SLX_ENODEVICE


def test_opt_init_matches_oracle(ffi, orc):
    o = ffi.Opt()
    ffi.lib().slx_opt_init(C.byref(o))
    e = orc.default_opt()
    for name, _ in ffi.Opt._fields_:
        a, b = getattr(o, name), getattr(e, name)
        assert (list(a) == list(b)) if name == "mat" else (a == b), name
    m1, m2 = (C.c_int8 * 25)(), (C.c_int8 * 25)()
    ffi.lib().slx_fill_scmat(2, 18, m1)
    orc.lib().orc_fill_scmat(2, 18, m2)
    assert list(m1) == list(m2)


def test_index_files_roundtrip_through_c_abi(ffi, golden_dir, tmp_path):
    """slx_index_load reads the reference's own `bwa index` fixture; slx_index_write reproduces it byte for byte"""
    import seqlib_amd
    idx = seqlib_amd.BWAIndex()
    assert idx.IsEmpty() and idx.NumSequences() == 0 and idx.printSamHeader() == ""
    with pytest.raises(RuntimeError):
        idx.ChrIDToName(0)
    with pytest.raises(RuntimeError):
        idx.WriteIndex(str(tmp_path / "x"))
    with pytest.raises(RuntimeError):
        idx.LoadIndex(str(tmp_path / "does_not_exist"))
    idx.LoadIndex(os.path.join(golden_dir, "tiny.fa"))
    assert idx.NumSequences() == 4 and [idx.ChrIDToName(i) for i in range(4)] == ["bcr", "abl", "tp53", "myc"]
    with pytest.raises(IndexError):
        idx.ChrIDToName(4)
    assert idx.printSamHeader() == "@SQ\tSN:bcr\tLN:141530\n@SQ\tSN:abl\tLN:178633\n@SQ\tSN:tp53\tLN:23070\n@SQ\tSN:myc\tLN:11518\n"
    assert str(idx) == "[BWAIndex] #seqs=4 pac_len=354751 holes=0"
    idx.WriteIndex(str(tmp_path / "w"))
    for ext in ("bwt", "sa", "pac", "ann", "amb"):
        assert filecmp.cmp(str(tmp_path / ("w." + ext)), os.path.join(golden_dir, "tiny.fa." + ext), shallow=False), ext
    with pytest.raises(RuntimeError):
        idx.WriteIndex(str(tmp_path))          # prefix is a directory (src/BWAIndex.cpp:393-395)
    # truncated file is rejected
    bad = tmp_path / "bad"
    for ext in ("bwt", "sa", "pac", "ann", "amb"):
        data = open(os.path.join(golden_dir, "tiny.fa." + ext), "rb").read()
        open(str(bad) + "." + ext, "wb").write(data[:len(data) // 2] if ext == "sa" else data)
    with pytest.raises(RuntimeError):
        seqlib_amd.BWAIndex().LoadIndex(str(bad))


def test_lrand48_helpers(ffi, orc):
    L = ffi.lib()
    libc = C.CDLL(None)
    libc.lrand48.restype = C.c_long
    libc.srand48(777)
    st = L.slx_lrand48_peek_libc()
    assert st == ((777 << 16) | 0x330E)
    assert L.slx_lrand48_peek_libc() == st                        # peeking does not disturb the stream
    want = [libc.lrand48() for _ in range(5)]
    assert [orc.lib().orc_lrand48_nth(st, k + 1) for k in range(5)] == want
    assert L.slx_lrand48_advance(st, 5) == L.slx_lrand48_peek_libc()
    libc.srand48(777)
    L.slx_lrand48_skip_libc(3)
    assert libc.lrand48() == want[3]


def test_no_gpu_fails_loudly(ffi, golden_dir):
    """Without a HIP device the alignment path must refuse to run (no silent CPU fallback)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import seqlib_amd
    idx = seqlib_amd.BWAIndex()
    idx.LoadIndex(os.path.join(golden_dir, "tiny.fa"))
    al = seqlib_amd.BWAAligner(idx)
    with pytest.raises(ffi.SlxError) as e:
        al.alignSequences(["ACGT" * 30])
    assert e.value.code == ffi.SLX_ENODEVICE
    with pytest.raises(ffi.SlxError) as e:
        seqlib_amd.BWAIndex().ConstructIndex([("a", "ACGTACGTAC")])
    assert e.value.code == ffi.SLX_ENODEVICE
    with pytest.raises(ValueError):
        seqlib_amd.BWAIndex().ConstructIndex([("a", "ACGT"), ("", "ACGT")])


def test_product_does_not_touch_the_oracle():
    """Nothing under seqlib_amd/ or include/ may include, import, link or load anything from oracle/."""
    bad = []
    for base in ("seqlib_amd", "include"):
        for dp, _, fns in os.walk(os.path.join(ROOT, base)):
            if "build" in dp.split(os.sep) or "__pycache__" in dp:
                continue
            for fn in fns:
                if not fn.endswith((".py", ".h", ".hip", ".cpp", ".c")):
                    continue
                txt = open(os.path.join(dp, fn), errors="replace").read()
                if re.search(r"oracle[/.]|liborc|orc_[a-z]|from oracle|import oracle", txt):
                    bad.append(os.path.join(dp, fn))
    assert not bad, bad


def test_library_load_leaves_the_environment_alone():
    """several aligners / fml contexts side by side want more than the runtime's four hardware queues ("Hardware queues"), but
    GPU_MAX_HW_QUEUES is read once per PROCESS by the HIP runtime and belongs to the application: loading the library, or importing the Python
    package, must not set it -- checked in a fresh interpreter on the C environment (os.environ is Python's copy from start-up)"""
    import subprocess, sys
    from seqlib_amd import _ffi
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import ctypes, sys; sys.path.insert(0, %r); ctypes.CDLL(%r); import seqlib_amd; c = ctypes.CDLL(None); c.getenv.restype = ctypes.c_char_p; "
            "v = c.getenv(b'GPU_MAX_HW_QUEUES'); print(v.decode() if v else None)") % (root, _ffi.SO_PATH)
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    assert subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True).stdout.strip() == "None"
    env["GPU_MAX_HW_QUEUES"] = "16"
    assert subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True).stdout.strip() == "16"


def test_nm8_chunk_against_the_definition(tmp_path):
    """k_cig_fast_coop's arithmetic (seqlib_amd/csrc/dev_nm8.h: mismatches of up to eight bases from one 8-byte read of the query and one of the packed
    text) against bns_get_seq's base at every coordinate: every start, length 1..8, both strands, both ends of the text -- under ASan + UBSan with the text
    padded by exactly the eight bytes the function may read past its end."""
    src = os.path.join(ROOT, "tests", "cpp", "nm8_test.cpp")
    exe = str(tmp_path / "nm8_test")
    subprocess.check_call(["g++", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-Wno-unknown-pragmas", "-DNM8_PAD=8", "-o", exe, src])
    out = subprocess.run([exe], stdout=subprocess.PIPE, check=True).stdout.decode()
    assert out.strip().endswith(" 0 bad"), out
