"""Build recipe for libtacotron2_amd.so (hand-written gfx950 HIP kernels behind a C ABI, include/tacotron2_amd.h).

hipcc cross-compiles for gfx950 without a GPU, so this runs in the build container; the resulting .so sits
in-tree (git-ignored) and travels to the GPU box with the snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libtacotron2_amd.so")
SOURCES = ["t2_error.cpp", "t2_gemm.hip", "t2_lstm.hip", "t2_attention.hip", "t2_elementwise.hip", "t2_logmel.hip",
           "t2_infer.hip"]
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]


def _stale(lib: str = LIB) -> bool:
    """True when `lib` is older than any kernel source, the ABI header or this recipe (its flags)."""
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "tacotron2_amd.h"),
                                                               os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


STAMPS_LIB = os.path.join(HERE, "libtacotron2_amd_stamps.so")


def build_variant(tag: str, defines, verbose: bool = True) -> str:
    """A DIAGNOSTIC library next to the product one: the same sources with extra -D defines, as libtacotron2_amd_<tag>.so (its
    own object directory; same dependency list as the product library: sources, ABI header, this recipe).  Select it with
    T2_LIB_PATH (tacotron2_amd/_lib.py, which checks the struct layouts of whatever it loads)."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    lib = os.path.join(HERE, f"libtacotron2_amd_{tag}.so")
    if not _stale(lib):
        return lib
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    objdir = os.path.join(HERE, "..", "build", f"obj_{tag}")
    os.makedirs(objdir, exist_ok=True)
    procs, objs = [], []
    for s in srcs:
        o = os.path.join(objdir, os.path.basename(s) + ".o")
        objs.append(o)
        cmd = [hipcc] + FLAGS + [f"-D{d}" for d in defines] + (["-x", "hip"] if s.endswith(".cpp") else []) + ["-c", s, "-o", o]
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {s}:\n{out}")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    if verbose:
        print(f"built {lib} (diagnostic: {' '.join('-D' + d for d in defines)})")
    return lib


def build_stamps(verbose: bool = True) -> str:
    """The diagnostic library with -DT2_STAMPS, i.e. with the in-kernel phase stamps (s_memtime words of one workgroup, read by
    tools/ubench_attn.py and tools/stamps_bwd.py) compiled in.  The product library has none: each stamp is a branch that ends a
    basic block, and the kernels of the frame chains run 0.3-0.5 us longer per launch with them."""
    return build_variant("stamps", ["T2_STAMPS"], verbose)


def build(force: bool = False, verbose: bool = True) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not force and not _stale():
        return LIB
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    objdir = os.path.join(HERE, "..", "build", "obj")
    os.makedirs(objdir, exist_ok=True)
    objs = []
    procs = []
    for s in srcs:
        o = os.path.join(objdir, os.path.basename(s) + ".o")
        objs.append(o)
        hdrs = [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".hpp")]
        if not force and os.path.exists(o) and os.path.getmtime(o) > max(
                [os.path.getmtime(s), os.path.getmtime(os.path.join(HERE, "..", "include", "tacotron2_amd.h")),
                 os.path.getmtime(os.path.abspath(__file__))] +
                [os.path.getmtime(h) for h in hdrs]):
            continue
        cmd = [hipcc] + FLAGS + (["-x", "hip"] if s.endswith(".cpp") else []) + ["-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {s}:\n{out}")
        if verbose and out.strip():
            print(out)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    if "--stamps" in sys.argv:
        print(build_stamps())
    elif "--variant" in sys.argv:          # python -m tacotron2_amd.build --variant TAG DEFINE[=VALUE] ...
        i = sys.argv.index("--variant")
        print(build_variant(sys.argv[i + 1], sys.argv[i + 2:]))
    else:
        build(force="--force" in sys.argv)
        print(LIB)
