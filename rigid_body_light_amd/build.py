"""In-tree build of the native parts (no network, no cmake needed):

  librbl.so            hipcc --offload-arch=gfx950   csrc/rbl_*.hip + rbl_host.cpp
  c_rigid.<abi>.so     g++ + pybind11               csrc/c_rigid.cpp  (links librbl.so, rpath $ORIGIN)

Both land next to this file so they travel with the tree (gpurun snapshot) and are the
files the Python package loads.  `python rigid_body_light_amd/build.py [--force]`.
"""
import os
import subprocess
import sys
import sysconfig
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
ARCH = os.environ.get("RBL_OFFLOAD_ARCH", "gfx950")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

HIP_SOURCES = ["rbl_kernels.hip", "rbl_dense.hip", "rbl_tilechol.hip", "rbl_body_dev.hip", "rbl_small.hip", "rbl_core.hip", "rbl_options.hip", "rbl_comm.hip",
               "rbl_products.hip", "rbl_bodies.hip", "rbl_roots.hip", "rbl_solvers.hip", "rbl_steps.hip", "rbl_host.cpp"]
HEADERS = ["rbl_internal.hpp", "rbl_api_internal.hpp", "rbl_pair.hpp", "rbl_pair_pk.hpp", "rbl_dense_dev.hpp", os.path.join("..", "..", "include", "rbl.h")]


def lib_path():
    return os.path.join(HERE, "librbl.so")


def isa_path():
    return os.path.join(HERE, "librbl.isa.json")


def ext_path():
    return os.path.join(HERE, "c_rigid" + sysconfig.get_config_var("EXT_SUFFIX"))


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _run(cmd):
    print("+", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def build(force=False, verbose_resources=False):
    os.makedirs(OBJ, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs, jobs = [], []
    for s in HIP_SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, s.rsplit(".", 1)[0] + ".o")
        objs.append(obj)
        if force or _newer(obj, [src] + hdrs):
            cmd = [HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=" + ARCH, "-x", "hip",
                   "-Wall", "-Wno-unused-function"] + os.environ.get("RBL_EXTRA_FLAGS", "").split() + ["-c", src, "-o", obj]
            if verbose_resources:
                cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
            jobs.append(cmd)
    if jobs:
        with ThreadPoolExecutor(max_workers=4) as ex:
            list(ex.map(_run, jobs))
    if force or jobs or _newer(lib_path(), objs):
        _run([HIPCC, "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", lib_path()] + objs)
    # executed-instruction counts of the matvec kernels (bench.py prices its roofline with them): the gfx950
    # assembly of the same source with the same flags, analysed by tools/isa_stats.py
    ksrc = os.path.join(CSRC, "rbl_kernels.hip")
    tool = os.path.join(HERE, "..", "tools", "isa_stats.py")
    if force or _newer(isa_path(), [ksrc, tool] + hdrs):
        asm = os.path.join(OBJ, "rbl_kernels.s")
        _run([HIPCC, "-O3", "-std=c++17", "--offload-arch=" + ARCH, "-x", "hip", "--offload-device-only", "-S",
              "-Wno-unused-command-line-argument"] + os.environ.get("RBL_EXTRA_FLAGS", "").split() + [ksrc, "-o", asm])
        _run([sys.executable, tool, asm, isa_path()])
    ext_src = os.path.join(CSRC, "c_rigid.cpp")
    if force or _newer(ext_path(), [ext_src, lib_path(), hdrs[-1]]):
        import pybind11
        _run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden",
              "-I" + pybind11.get_include(), "-I" + sysconfig.get_paths()["include"],
              ext_src, "-o", ext_path(), "-L" + HERE, "-lrbl", "-Wl,-rpath,$ORIGIN"])
    # the C++-only multi-GPU host (examples/host_rccl_step.cpp): librbl's C ABI and nothing else
    host_src = os.path.join(HERE, "..", "examples", "host_rccl_step.cpp")
    host_bin = os.path.join(HERE, "..", "examples", "host_rccl_step")
    if os.path.exists(host_src) and (force or _newer(host_bin, [host_src, lib_path(), hdrs[-1]])):
        _run([HIPCC, "-O2", "-std=c++17", host_src, "-I" + os.path.join(HERE, "..", "include"), "-L" + HERE, "-lrbl",
              "-Wl,-rpath,$ORIGIN/../rigid_body_light_amd", "-o", host_bin])
    return lib_path(), ext_path()


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose_resources="--resources" in sys.argv)
