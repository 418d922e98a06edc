"""In-tree build of the native parts (no cmake/ninja needed):

  libf2n_hip.so   the C-ABI library (include/f2n_abi.h): csrc/*.hip compiled by hipcc for gfx950
  _f2n_host*.so   the C++/LibTorch host layer (csrc/host/*.cpp, pybind11 module) linked against it

Both land next to this file so that they travel with a `gpurun` snapshot.  `-ffp-contract=off` is part of the
numerical contract with the oracle (integer outputs of the sampler / hash grid depend on the fp32 op order).
"""
import concurrent.futures
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
OBJ = os.path.join(HERE, "build")

# Two builds of the same sources (include/f2n_abi.h: f2n_numerics_mode):
#   ""        the product: libf2n_hip.so + _f2n_host*.so
#   "refnum"  -DF2N_REFERENCE_NUMERICS=1: libf2n_hip_refnum.so + _f2n_host_refnum*.so -- the reference's per-addend f16 hash
#             gradient atomics and an f16 MLP forward accumulator, for A/B trainings (bench.py psnr_numerics_ab).  A process
#             picks it with F2N_REFERENCE_NUMERICS=1 in its environment BEFORE the package is imported (one numerics per process).
#   "debug"   -DF2N_DEBUG_BUILD=1: libf2n_hip_debug.so + _f2n_host_debug*.so -- the product's code plus the debugging launches of
#             include/f2n_debug.h, the stream-skew hooks of the host's Renderer and the measurement knobs that read the
#             environment (the product library reads none).  Picked with F2N_DEBUG_BUILD=1 in the environment before the import.
VARIANT = ("refnum" if os.environ.get("F2N_REFERENCE_NUMERICS", "0") not in ("", "0") else
           "debug" if os.environ.get("F2N_DEBUG_BUILD", "0") not in ("", "0") else "")


def lib_path(variant=None):
    v = VARIANT if variant is None else variant
    return os.path.join(HERE, "libf2n_hip%s.so" % ("_" + v if v else ""))


LIB = lib_path()

HIP_SOURCES = ["sampler.hip", "field.hip", "shade.hip", "render.hip", "optim.hip", "workspace.hip", "dataset.hip", "octree.hip",
               "mlp_generic.hip"]
HIP_HEADERS = ["f2n_dev.h", "mlp_dev.h", "rows_dev.h", "adam_dev.h", os.path.join(INCLUDE, "f2n_abi.h"), os.path.join(INCLUDE, "f2n_debug.h")]
# (-mllvm -amdgpu-mfma-vgpr-form=1 was tried: a third fewer instructions in the MLP backward kernels -- no
# v_accvgpr_read of every MFMA result -- but 15 % SLOWER: those kernels are bound by dependency latency, not issue.)
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
               "-Wall", "-Wno-unused-function"]


def _hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_hip(force=False, verbose=False, variant=None):
    variant = VARIANT if variant is None else variant
    lib = lib_path(variant)
    os.makedirs(OBJ, exist_ok=True)
    hdrs = [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HIP_HEADERS]
    jobs = []
    objs = []
    defs = {"refnum": ["-DF2N_REFERENCE_NUMERICS=1"], "debug": ["-DF2N_DEBUG_BUILD=1"]}.get(variant, [])
    extra = os.environ.get("F2N_EXTRA_HIPCC", "")  # measurement knob: "file.hip:-flag -flag;other.hip:-flag"
    extra_by_file = dict(kv.split(":", 1) for kv in extra.split(";") if ":" in kv)
    for src in HIP_SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".hip", (".%s.o" % variant) if variant else ".o"))
        objs.append(o)
        if force or _newer(o, [s] + hdrs) or src in extra_by_file:
            jobs.append([_hipcc()] + HIPCC_FLAGS + defs + extra_by_file.get(src, "").split() + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n%s\n%s" % (" ".join(cmd), r.stderr[-6000:]))
        return r.stderr

    with concurrent.futures.ThreadPoolExecutor(max_workers=6) as ex:
        for warn in ex.map(run, jobs):
            if verbose and warn.strip():
                print(warn[-3000:])
    if force or jobs or _newer(lib, objs):
        run([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", lib])
    return lib


def host_module_path(variant=None):
    import sysconfig
    v = VARIANT if variant is None else variant
    return os.path.join(HERE, "_f2n_host" + ("_" + v if v else "") + sysconfig.get_config_var("EXT_SUFFIX"))


def build_host(force=False, verbose=False, variant=None):
    """C++/LibTorch host layer as a pybind11 extension, compiled with g++ against the pip wheel's headers."""
    import sysconfig
    import torch
    from torch.utils import cpp_extension as ce
    host_dir = os.path.join(CSRC, "host")
    srcs = sorted(os.path.join(host_dir, f) for f in os.listdir(host_dir) if f.endswith(".cpp"))
    hdrs = [os.path.join(host_dir, f) for f in os.listdir(host_dir) if f.endswith(".h")] + [os.path.join(INCLUDE, "f2n_abi.h"),
                                                                                             os.path.join(INCLUDE, "f2n_debug.h")]
    variant = VARIANT if variant is None else variant
    out = host_module_path(variant)  # (product and refnum: the same objects, only the kernel library linked against differs;
    #                                     debug: its own objects, compiled with -DF2N_DEBUG_BUILD=1)
    dbg = variant == "debug"
    os.makedirs(OBJ, exist_ok=True)
    inc = ["-I" + p for p in ce.include_paths()] + ["-I" + sysconfig.get_paths()["include"], "-I" + INCLUDE,
                                                      "-I/opt/rocm/include"]
    cxx11 = int(torch._C._GLIBCXX_USE_CXX11_ABI)
    flags = ["-O2", "-std=c++17", "-fPIC", "-DTORCH_EXTENSION_NAME=_f2n_host", "-DTORCH_API_INCLUDE_EXTENSION_H",
             "-D_GLIBCXX_USE_CXX11_ABI=%d" % cxx11, "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-w"] + (["-DF2N_DEBUG_BUILD=1"] if dbg else [])
    jobs, objs = [], []
    for s in srcs:
        o = os.path.join(OBJ, "host_" + os.path.basename(s).replace(".cpp", ".debug.o" if dbg else ".o"))
        objs.append(o)
        if force or _newer(o, [s] + hdrs):
            jobs.append(["g++"] + flags + inc + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd[:8]), "...", cmd[-3], flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("g++ failed:\n%s\n%s" % (" ".join(cmd), r.stderr[-8000:]))

    with concurrent.futures.ThreadPoolExecutor(max_workers=8) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _newer(out, objs + [lib_path(variant)]):
        libdir = os.path.join(os.path.dirname(torch.__file__), "lib")
        link = ["g++", "-shared"] + objs + ["-o", out, "-L" + libdir, "-L" + HERE, "-Wl,-rpath," + libdir,
                                            "-Wl,-rpath,$ORIGIN", "-lf2n_hip" + ("_" + variant if variant else ""), "-lrccl", "-lc10", "-ltorch_cpu", "-ltorch",
                                            "-ltorch_python", "-lc10_hip", "-ltorch_hip"]
        run(link)
    return out


def build_all(force=False, verbose=False, variants=("", "refnum", "debug")):
    """Builds every variant; returns the paths of the one this process uses (VARIANT)."""
    out = {}
    for v in variants:
        lib = build_hip(force, verbose, v)
        host = None
        if os.path.isdir(os.path.join(CSRC, "host")):
            host = build_host(force, verbose, v)
        out[v] = (lib, host)
    return out.get(VARIANT, out[variants[0]])


if __name__ == "__main__":
    print(build_all(force="--force" in sys.argv, verbose=True))
