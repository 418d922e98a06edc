"""Static resource table of every gfx950 kernel in the product library, read from the code objects hipcc has just built
(f2-nerf_amd/build/*.o): registers, LDS, scratch, spills, and the resident waves per SIMD those imply.  No GPU needed.

  python tools/kernel_resources.py [--variant ''|refnum|debug] [--out profiles/rNN_kernel_resources.txt]

What it is for: the occupancy figures DESIGN.md section 3/4 argue from (254-register backward kernels at two blocks per CU,
weights in LDS taking field_shade_fwd from 152 to 80 registers, ...) can be read off the build instead of being retold, and
tests/test_abi_cpu.py checks on every CPU run that no kernel of the product spills or touches scratch memory."""
import argparse
import glob
import os
import subprocess
import sys
import tempfile

import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"


def code_object_metadata(obj_path):
    """amdhsa.kernels of the gfx950 code object embedded in one host object file (hipcc -c output)."""
    with tempfile.TemporaryDirectory() as tmp:
        fat, co = os.path.join(tmp, "fatbin"), os.path.join(tmp, "co")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", obj_path, fat], check=True)
        subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                        "--targets=" + TARGET, "--output=" + co], check=True)
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], check=True, capture_output=True, text=True).stdout
    start = notes.index("---")
    end = notes.index("\n...", start)
    return yaml.safe_load(notes[start:end])["amdhsa.kernels"]


def _itanium_head(name):
    """kernel name + integer / bool template arguments of an Itanium-mangled function, for the names c++filt gives up on (it does not
    know DF16_, the mangling of _Float16): _Z16field_bwd_kernelILi1ELi0ELi3EEv... -> field_bwd_kernel<1, 0, 3>"""
    if not name.startswith("_Z"):
        return name
    i = 2
    while i < len(name) and name[i].isdigit():
        i += 1
    n = int(name[2:i])
    head, rest = name[i:i + n], name[i + n:]
    args = []
    if rest.startswith("I"):
        j = 1
        while j < len(rest) and rest[j] == "L":
            k = rest.index("E", j)
            kind, val = rest[j + 1], rest[j + 2:k]
            args.append({"0": "false", "1": "true"}.get(val, val) if kind == "b" else val)
            j = k + 1
    return head + ("<%s>" % ", ".join(args) if args else "")


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.splitlines()
    return [_itanium_head(n) if o == n else o for n, o in zip(names, out)]


def short(name):
    """kernel<template args> without the argument list"""
    depth, cut = 0, len(name)
    for i, ch in enumerate(name):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            cut = i
            break
    s = name[:cut]
    return s[5:] if s.startswith("void ") else s


def waves_per_simd(vgpr, agpr, lds, wg_size):
    """Resident waves per SIMD on gfx950: 512 unified VGPRs per lane and SIMD (allocation granule 8, arch + acc), at most 8 waves;
    160 KB of LDS per CU, 4 SIMDs per CU."""
    regs = max(8, -(-(vgpr + agpr) // 8) * 8)
    by_regs = min(8, 512 // regs)
    if lds <= 0:
        return by_regs, by_regs, None
    waves_per_wg = max(1, -(-wg_size // 64))
    wgs_by_lds = (160 * 1024) // lds
    by_lds = wgs_by_lds * waves_per_wg / 4.0
    return min(by_regs, by_lds), by_regs, by_lds


def table(variant=""):
    rows = []
    suffix = (".%s.o" % variant) if variant else ".o"
    for obj in sorted(glob.glob(os.path.join(ROOT, "f2-nerf_amd", "build", "*" + suffix))):
        base = os.path.basename(obj)
        if base.startswith("host_") or (not variant and base.count(".") != 1):
            continue
        kernels = code_object_metadata(obj)
        names = demangle([k[".name"] for k in kernels])
        for k, n in zip(kernels, names):
            w, by_regs, by_lds = waves_per_simd(k[".vgpr_count"], k.get(".agpr_count", 0), k[".group_segment_fixed_size"],
                                                k[".max_flat_workgroup_size"])
            rows.append({"file": base[:-len(suffix)] + ".hip", "kernel": short(n), "vgpr": k[".vgpr_count"], "agpr": k.get(".agpr_count", 0),
                         "sgpr": k[".sgpr_count"], "lds": k[".group_segment_fixed_size"], "scratch": k[".private_segment_fixed_size"],
                         "vgpr_spill": k.get(".vgpr_spill_count", 0), "sgpr_spill": k.get(".sgpr_spill_count", 0),
                         "wg": k[".max_flat_workgroup_size"], "waves_per_simd": w, "dyn_lds": bool(k.get(".uses_dynamic_stack", False))})
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variant", default="")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    rows = table(a.variant)
    lines = ["# static resources of the gfx950 kernels in libf2n_hip%s.so (tools/kernel_resources.py; hipcc metadata, no GPU)" % (("_" + a.variant) if a.variant else ""),
             "# waves/SIMD: min(8, 512 // ceil8(vgpr + agpr)) capped by STATIC LDS (160 KB per CU, 4 SIMDs); kernels that also take dynamic",
             "# LDS at launch (shade_bwd: the embedding image; hash_bin_accumulate: the fp64 slice image) sit lower than listed",
             "%-14s %-58s %5s %5s %5s %7s %7s %6s %5s %6s" % ("file", "kernel", "vgpr", "agpr", "sgpr", "lds", "scratch", "spills", "wg", "waves")]
    for r in sorted(rows, key=lambda r: (r["file"], r["kernel"])):
        lines.append("%-14s %-58s %5d %5d %5d %7d %7d %6d %5d %6.1f" % (r["file"], r["kernel"][:58], r["vgpr"], r["agpr"], r["sgpr"], r["lds"],
                                                                  r["scratch"], r["vgpr_spill"] + r["sgpr_spill"], r["wg"], r["waves_per_simd"]))
    n_bad = sum(1 for r in rows if r["scratch"] or r["vgpr_spill"] or r["sgpr_spill"])
    lines.append("# %d kernels; %d with scratch memory or spills" % (len(rows), n_bad))
    text = "\n".join(lines) + "\n"
    if a.out:
        with open(a.out, "w") as f:
            f.write(text)
    sys.stdout.write(text)


if __name__ == "__main__":
    main()
