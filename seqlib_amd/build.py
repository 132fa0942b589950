"""Builds libseqlib_amd.so (HIP kernels + C-ABI) in-tree for gfx950 with hipcc."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(HERE, "libseqlib_amd.so")
SOURCES = ["slx_index.cpp", "slx_index_gpu.hip", "slx_index_gpu64.hip", "slx_align.hip", "slx_align_wide.hip", "slx_fml.hip", "slx_fml_asm.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wno-unused-value",
         "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # two groups of translation units with their own headers: the BWAAligner path, and the FermiAssembler / BFC path (slx_fml*)
    is_fml = lambda f: "fml" in f
    hdr_align = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc")) and not is_fml(f)] + [os.path.join(ROOT, "include", "seqlib_amd.h")]
    hdr_fml = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h") and is_fml(f)] + \
              [os.path.join(CSRC, "slx_internal.h"), os.path.join(ROOT, "include", "seqlib_amd.h"), os.path.join(ROOT, "include", "seqlib_amd_fml.h")]
    objs, procs = [], []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    for s in SOURCES:                      # the translation units compile side by side
        src = os.path.join(CSRC, s)
        obj = os.path.join(HERE, "build", s + ".o")
        if force or _stale(obj, [src] + (hdr_fml if is_fml(s) else hdr_align)):
            cmd = [hipcc] + FLAGS + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            procs.append((cmd, subprocess.Popen(cmd)))
        objs.append(obj)
    for cmd, p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    if force or _stale(SO, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", SO]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    # tools/*: the C++ drop-in class timed end to end (bench.py's value_bamrecords, value_per_call); host-only code over the C-ABI
    hdrs = [os.path.join(ROOT, "include", "SeqLib", f) for f in os.listdir(os.path.join(ROOT, "include", "SeqLib"))]
    for name in ("bamrec_bench", "percall_bench"):
        tool_src = os.path.join(ROOT, "tools", name + ".cpp")
        tool = os.path.join(HERE, name)
        if force or _stale(tool, [tool_src, SO] + hdrs):
            cmd = ["g++", "-std=c++17", "-O2", "-I" + os.path.join(ROOT, "include"), tool_src, "-o", tool, "-L" + HERE, "-lseqlib_amd",
                   "-Wl,-rpath," + HERE, "-Wl,-rpath,/opt/rocm/lib", "-lz", "-lpthread"]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.check_call(cmd)
    return SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
