"""Build the gfx950 shared library (hipcc cross-compiles without a GPU).

    python -m zhilight_amd.build [--force]

Produces zhilight_amd/libzhilight_amd.so in-tree (git-ignored, but it travels with gpurun snapshots).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libzhilight_amd.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-fno-fast-math", "-ffp-contract=off"]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    srcs = sources()
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(HERE, "..", "include", "zhilight_amd.h"))
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    objs = []
    jobs = []
    for s in srcs:
        o = os.path.join(objdir, os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            jobs.append([HIPCC] + FLAGS + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if jobs or force or _stale(OUT, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs)
    build_hostcpp(force, verbose)
    build_comm(force, verbose)
    return OUT


COMM_OUT = os.path.join(HERE, "libzhilight_amd_comm.so")


def build_comm(force=False, verbose=False):
    """libzhilight_amd_comm.so: the tensor-parallel exchange step (direct RCCL communicator + one-shot peer-read all-reduce,
    include/zhilight_amd_comm.h).  Separate from the main library so that single-GPU users carry no RCCL dependency."""
    src = os.path.join(HERE, "csrc_comm", "comm.hip")
    deps = [src, os.path.join(HERE, "..", "include", "zhilight_amd_comm.h"), os.path.join(HERE, "..", "include", "zhilight_amd.h")]
    if not (force or _stale(COMM_OUT, deps)):
        return COMM_OUT
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-shared", "-o", COMM_OUT, src,
           "-L" + os.path.join(rocm, "lib"), "-lrccl"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return COMM_OUT


HOSTCPP = os.path.join(HERE, "hostcpp")


def hostcpp_target():
    import sysconfig
    return os.path.join(HERE, "zl_internals" + sysconfig.get_config_var("EXT_SUFFIX"))


def build_hostcpp(force=False, verbose=False):
    """The C++ host layer (bmengine-on-HIP shim + the reference's nn:: / int8_op:: operator names over the C ABI) and its
    pybind11 test module zhilight_amd/zl_internals*.so.  Host code only: g++ against the HIP runtime headers."""
    import pybind11
    import sysconfig
    srcs = [os.path.join(HOSTCPP, f) for f in ("bm_hip.cpp", "nn_amd.cpp", "py_internals.cpp")]
    deps = srcs + [os.path.join(HOSTCPP, f) for f in ("bm_hip.h", "nn_amd.h")] + [os.path.join(HERE, "..", "include", "zhilight_amd.h")]
    target = hostcpp_target()
    if not (force or _stale(target, deps)):
        return target
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(rocm, "include"),
           "-I" + pybind11.get_include(), "-I" + sysconfig.get_paths()["include"]] + srcs + [
           "-o", target, "-L" + HERE, "-lzhilight_amd", "-L" + os.path.join(rocm, "lib"), "-lamdhip64",
           "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + os.path.join(rocm, "lib")]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return target


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
