"""Build liblrx.so (HIP kernels + C ABI) in-tree for gfx950 with hipcc.  No torch involved."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.environ.get("LRX_CSRC_DIR") or os.path.join(HERE, "csrc")   # env override: tools/ diagnostic variants build from a patched copy
LIB = os.path.join(HERE, "liblrx.so")
SOURCES = ["lrx_capi.hip", "lrx_elementwise.hip", "lrx_gemm.hip", "lrx_attn.hip", "lrx_search.hip", "lrx_sparse.hip", "lrx_fuse.hip"]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "lrx.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True, defines=(), out: str = None) -> str:
    """defines/out: diagnostic variant builds (tools/ only); the product library is always LIB."""
    if out is None and not force and not _stale():
        return LIB
    lib_path = out or LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    for src in SOURCES:
        obj = os.path.join(HERE, "build", src.replace(".hip", ".o"))
        obj = obj if out is None else obj.replace(".o", "." + os.path.basename(out) + ".o")
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"] + ["-D" + d for d in defines] + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print("[lrx build]", " ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for src, p in procs:
        log, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(log.decode())
            raise RuntimeError(f"hipcc failed on {src}")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_path] + objs
    if verbose:
        print("[lrx build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return lib_path


TORCH_LIB = os.path.join(HERE, "liblrx_torch.so")


def build_torch_ops(force: bool = False, verbose: bool = True) -> str:
    """liblrx_torch.so: the TORCH_LIBRARY(lrx, ...) registration layer over liblrx.so (csrc/lrx_torch.cpp), compiled in-tree against the
    installed PyTorch-ROCm headers.  Plain host C++ (no device code): the kernels stay in liblrx.so."""
    src = os.path.join(CSRC, "lrx_torch.cpp")
    deps = [src, os.path.join(HERE, "..", "include", "lrx.h")]
    if not force and os.path.exists(TORCH_LIB) and all(os.path.getmtime(d) <= os.path.getmtime(TORCH_LIB) for d in deps) and os.path.exists(LIB):
        return TORCH_LIB
    build(force=False, verbose=verbose)
    import torch
    from torch.utils import cpp_extension as ce
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    cmd = [hipcc, "-x", "c++", "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI)]
    cmd += ["-I" + p for p in ce.include_paths()] + ["-I/opt/rocm/include", src, "-o", TORCH_LIB, "-L" + tlib, "-L" + HERE, "-llrx", "-ltorch", "-ltorch_cpu",
                                                    "-ltorch_hip", "-lc10", "-lc10_hip", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + tlib]
    if verbose:
        print("[lrx build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return TORCH_LIB


if __name__ == "__main__":
    defs = [a[2:] for a in sys.argv[1:] if a.startswith("-D")]
    outs = [a[6:] for a in sys.argv[1:] if a.startswith("--out=")]
    print(build(force="--force" in sys.argv, defines=defs, out=outs[0] if outs else None))
    if not defs and not outs:
        print(build_torch_ops(force="--force" in sys.argv))
