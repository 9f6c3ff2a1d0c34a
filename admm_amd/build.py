"""Build recipe of libadmm_hip.so (hipcc, gfx950 only) and of the C++ host programs."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
HOST = os.path.join(PKG, "host")
OBJ = os.path.join(PKG, "csrc", "_obj")
LIB = os.path.join(PKG, "libadmm_hip.so")

HIP_SOURCES = ["ctx.hip", "csr.hip", "copies.hip", "forward_rows.hip", "forward_panels.hip", "transposed.hip", "solver.hip", "ftrl_batch.hip", "ftrl.hip",
               "auc.hip", "comm.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
HIP_FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-munsafe-fp-atomics", "-ffp-contract=off",
             "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result",
             "-I", os.path.join(ROOT, "include"), "-I", CSRC]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src):
    obj = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
    deps = [os.path.join(CSRC, src), os.path.join(CSRC, "internal.h"), os.path.join(CSRC, "sweep_common.h"),
            os.path.join(ROOT, "include", "admm_hip.h")]
    if _newer(obj, deps):
        subprocess.check_call([HIPCC] + HIP_FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj])
    return obj


def build_lib(verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    with ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 2)) as ex:
        objs = list(ex.map(_compile, HIP_SOURCES))
    if _newer(LIB, objs):
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-o", LIB] + objs + ["-ldl"])
    if verbose:
        print("built", LIB)
    return LIB


HOST_SOURCES = ["runtime.cc", "arg_parser.cc", "sample_set.cc", "solvers.cc", "model_io.cc"]
HOST_MAINS = {"admm": "admm_main.cc", "ftrl": "ftrl_main.cc", "evan": "eval_main.cc", "parsecheck": "parse_main.cc"}


def build_host(verbose=False):
    """g++ build of the reference-compatible host classes and the admm / ftrl / evan programs."""
    bindir = os.path.join(PKG, "bin")
    os.makedirs(bindir, exist_ok=True)
    cxx = os.environ.get("CXX", "g++")
    flags = ["-O2", "-std=c++17", "-pthread", "-Wall", "-I", os.path.join(ROOT, "include"), "-I", HOST]
    srcs = [os.path.join(HOST, s) for s in HOST_SOURCES if os.path.exists(os.path.join(HOST, s))]
    outs = []
    for name, main in HOST_MAINS.items():
        mp = os.path.join(HOST, main)
        if not os.path.exists(mp):
            continue
        out = os.path.join(bindir, name)
        hdrs = [os.path.join(HOST, h) for h in os.listdir(HOST) if h.endswith(".h")]
        if _newer(out, srcs + [mp, LIB] + hdrs):
            subprocess.check_call([cxx] + flags + [mp] + srcs +
                                  ["-L", PKG, "-ladmm_hip", "-Wl,-rpath,$ORIGIN/..", "-o", out])
        outs.append(out)
    if verbose:
        print("built", outs)
    return outs


if __name__ == "__main__":
    build_lib(verbose=True)
    build_host(verbose=True)
