"""Test infrastructure -- NOT part of the product.  Builds tests/wave_emul/_build/libf2n_emul.so: the product's kernel sources
(f2-nerf_amd/csrc/*.hip, read in place) compiled for x86-64 by ROCm's clang++ against the wavefront emulation of
tests/wave_emul/include/hip/hip_runtime.h.  The library exports the same C-ABI as libf2n_hip.so (include/f2n_abi.h), with host
pointers where the product takes device pointers.

The source text is compiled as it is, except for two spellings that only exist for the GPU target and are rewritten on the way
(the rewritten copies live in _build/, never in the tree; REWRITES lists them and the build reports how often each fired):
  * `extern __shared__ T name[];`      -> `T* name = (T*) wemu::dyn_lds();`   (dynamic LDS is the emulator's buffer)
  * `asm volatile("" : "+v"(x));`      -> `asm volatile("" : "+r"(x));`        (an optimisation barrier on a VGPR)
"""
import concurrent.futures
import os
import re
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "f2-nerf_amd", "csrc")
OUT = os.path.join(HERE, "_build")
LIB = os.path.join(OUT, "libf2n_emul.so")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"

SOURCES = ["sampler.hip", "render.hip", "dataset.hip", "octree.hip", "optim.hip", "workspace.hip", "field.hip", "shade.hip",
           "mlp_generic.hip"]
HEADERS = ["f2n_dev.h", "rows_dev.h", "mlp_dev.h", "adam_dev.h"]
REWRITES = [
    (re.compile(r"extern\s+__shared__\s+([A-Za-z_][A-Za-z_0-9 ]*?)\s+([A-Za-z_][A-Za-z_0-9]*)\s*\[\s*\]\s*;"),
     r"\1* \2 = (\1*) wemu::dyn_lds();"),
    (re.compile(r'asm\s+volatile\(""\s*:\s*"\+v"\(([A-Za-z_0-9]+)\)\);'), r'asm volatile("" : "+r"(\1));'),
]
# -ffp-contract=off: as the product build (f2-nerf_amd/build.py); binary16 arithmetic rounded after every operation, as the GPU's
# native f16 instructions do (clang's default for x86 keeps excess precision inside an expression)
FLAGS = ["-x", "c++", "-std=c++17", "-O0", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-Xclang", "-ffloat16-excess-precision=none",
         "-mf16c", "-w", "-I" + os.path.join(HERE, "include"), "-I" + OUT]


RT_FLAGS = [f if f != "-O0" else "-O2" for f in FLAGS] + ["-fno-omit-frame-pointer", "-mno-omit-leaf-frame-pointer", "-fno-optimize-sibling-calls"]  # (the emulation's own runtime)
# the kernels' basic blocks report to the runtime (wemu_rt.cpp: how it knows which lanes are behind)
FLAGS += ["-fno-omit-frame-pointer", "-fsanitize-coverage=trace-pc,no-prune"]


def _rewrite(text):
    fired = []
    for rx, to in REWRITES:
        text, n = rx.subn(to, text)
        fired.append(n)
    return text, fired


def build(sources=None, verbose=False, force=False, mutate=None, tag="", defines=(), traffic=False):
    """mutate = [(file name, old text, new text)]: a deliberately broken variant (built into _build/<tag>/) for the tests that show
    the emulated suite notices.  defines: e.g. ("F2N_REFERENCE_NUMERICS=1",), the second build of the product (f2-nerf_amd/build.py)."""
    sources = SOURCES if sources is None else sources
    out = os.path.join(OUT, tag) if tag else OUT
    lib = os.path.join(out, "libf2n_emul.so")
    os.makedirs(os.path.join(out, "csrc"), exist_ok=True)
    report = {}
    newest = 0.0
    for name in HEADERS + sources:
        src = os.path.join(CSRC, name)
        newest = max(newest, os.path.getmtime(src))
        with open(src) as f:
            text, fired = _rewrite(f.read())
        for mname, old, new in mutate or []:
            if mname == name:
                assert text.count(old) == 1, (name, old, text.count(old))
                text = text.replace(old, new)
        report[name] = fired
        dst = os.path.join(out, "csrc", name)
        if not os.path.exists(dst) or open(dst).read() != text:
            with open(dst, "w") as f:
                f.write(text)
    for dep in (os.path.join(HERE, "include", "hip", "hip_runtime.h"), os.path.join(HERE, "wemu_rt.cpp"), os.path.join(ROOT, "include", "f2n_abi.h"),
                os.path.abspath(__file__)):
        newest = max(newest, os.path.getmtime(dep))
    tagf = os.path.join(out, "sources.txt")
    want = " ".join(sources) + " | " + repr(mutate) + " | " + repr(tuple(defines)) + " | " + repr(traffic)
    if not force and os.path.exists(lib) and os.path.getmtime(lib) >= newest and os.path.exists(tagf) and open(tagf).read() == want:
        return lib, report
    # (csrc/f2n_dev.h includes "../../include/f2n_abi.h": resolved against the copies' directory first, then against -I paths --
    # tests/wave_emul/include/f2n_abi.h forwards to the repository's header)
    flags = FLAGS + ["-I" + os.path.join(HERE, "include", "hip", "..", "..")] + ["-D" + d for d in defines]
    if traffic:  # (every load and store of the kernels reports to the runtime as well: tests/wave_emul/source_level_traffic.py)
        flags = [f + ",trace-loads,trace-stores" if f.startswith("-fsanitize-coverage=") else f for f in flags]
    jobs = [[CLANG] + flags + ["-c", os.path.join(out, "csrc", s), "-o", os.path.join(out, s + ".o")] for s in sources]
    jobs.append([CLANG] + RT_FLAGS + ["-c", os.path.join(HERE, "wemu_rt.cpp"), "-o", os.path.join(out, "wemu_rt.o")])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("wave_emul build failed:\n%s\n%s" % (" ".join(cmd), r.stderr[-8000:]))

    with concurrent.futures.ThreadPoolExecutor(max_workers=8) as ex:
        list(ex.map(run, jobs))
    run([CLANG, "-shared", "-fPIC"] + [j[-1] for j in jobs] + ["-o", lib, "-lm"])
    with open(tagf, "w") as f:
        f.write(want)
    return lib, report


def build_selftest(force=False):
    """libwemu_selftest.so: tests/wave_emul/selftest.hip (known-answer kernels for the emulation itself) + the runtime."""
    os.makedirs(OUT, exist_ok=True)
    src = os.path.join(HERE, "selftest.hip")
    lib = os.path.join(OUT, "libwemu_selftest.so")
    deps = [src, os.path.join(HERE, "include", "hip", "hip_runtime.h"), os.path.join(HERE, "wemu_rt.cpp"), os.path.abspath(__file__)]
    if not force and os.path.exists(lib) and os.path.getmtime(lib) >= max(os.path.getmtime(d) for d in deps):
        return lib
    with open(src) as f:
        text, _ = _rewrite(f.read())
    with open(os.path.join(OUT, "selftest.hip"), "w") as f:
        f.write(text)
    for cmd in ([CLANG] + FLAGS + ["-c", os.path.join(OUT, "selftest.hip"), "-o", os.path.join(OUT, "selftest.o")],
                [CLANG] + RT_FLAGS + ["-c", os.path.join(HERE, "wemu_rt.cpp"), "-o", os.path.join(OUT, "wemu_rt.selftest.o")],
                [CLANG, "-shared", "-fPIC", os.path.join(OUT, "selftest.o"), os.path.join(OUT, "wemu_rt.selftest.o"), "-o", lib, "-lm"]):
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("wave_emul selftest build failed:\n" + r.stderr[-6000:])
    return lib


def build_host(force=False, verbose=False):
    """_build/_f2n_host_emul*.so: the C++/LibTorch host layer (f2-nerf_amd/csrc/host/*.cpp, read in place) compiled by g++ with
    host_shim/host_shim.h forced in front of every file and linked against libf2n_emul.so -- the plugin classes on CPU tensors."""
    import sysconfig
    import torch
    from torch.utils import cpp_extension as ce
    lib, _ = build()
    host_dir = os.path.join(CSRC, "host")
    srcs = sorted(os.path.join(host_dir, f) for f in os.listdir(host_dir) if f.endswith(".cpp"))
    shim = os.path.join(HERE, "host_shim")
    deps = srcs + [os.path.join(host_dir, f) for f in os.listdir(host_dir) if f.endswith(".h")] + [
        os.path.join(shim, "host_shim.h"), os.path.join(shim, "rccl", "rccl.h"), os.path.join(ROOT, "include", "f2n_abi.h"), os.path.abspath(__file__)]
    out = os.path.join(OUT, "_f2n_host_emul" + sysconfig.get_config_var("EXT_SUFFIX"))
    obj = os.path.join(OUT, "host")
    os.makedirs(obj, exist_ok=True)
    if not force and os.path.exists(out) and os.path.getmtime(out) >= max(max(os.path.getmtime(d) for d in deps), os.path.getmtime(lib)):
        return out
    inc = ["-I" + shim] + ["-I" + p for p in ce.include_paths()] + ["-I" + sysconfig.get_paths()["include"], "-I" + os.path.join(ROOT, "include"),
                                                                  "-I/opt/rocm/include"]
    flags = ["-O1", "-std=c++17", "-fPIC", "-DTORCH_EXTENSION_NAME=_f2n_host_emul", "-DTORCH_API_INCLUDE_EXTENSION_H",
             "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI), "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-w",
             # (its own pybind11 type registry: the product's host module may live in the same process and registers the same C++ classes)
             '-DPYBIND11_COMPILER_TYPE="_wemu"',
             "-include", os.path.join(shim, "host_shim.h")]
    jobs, objs = [], []
    for s in srcs:
        o = os.path.join(obj, os.path.basename(s).replace(".cpp", ".o"))
        objs.append(o)
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(d) for d in deps):
            jobs.append(["g++"] + flags + inc + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd[:6]), "...", cmd[-3], flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("wave_emul host build failed:\n%s\n%s" % (" ".join(cmd), r.stderr[-8000:]))

    with concurrent.futures.ThreadPoolExecutor(max_workers=8) as ex:
        list(ex.map(run, jobs))
    libdir = os.path.join(os.path.dirname(torch.__file__), "lib")
    run(["g++", "-shared"] + objs + ["-o", out, "-L" + libdir, "-L" + OUT, "-Wl,-rpath," + libdir, "-Wl,-rpath," + OUT, "-lf2n_emul", "-lc10", "-ltorch_cpu",
                                    "-ltorch", "-ltorch_python", "-lc10_hip", "-ltorch_hip"])
    return out


if __name__ == "__main__":
    import sys
    lib, rep = build(sources=sys.argv[1:] or None, verbose=True, force=True)
    print(lib)
    for k, v in rep.items():
        print("  %-18s dynamic-LDS declarations rewritten: %d, VGPR barriers rewritten: %d" % (k, v[0], v[1]))
