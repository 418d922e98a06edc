"""Static instruction mix of every gfx950 kernel in the product library (disassembly of the code objects hipcc has just built; no
GPU).  Counts are STATIC -- an instruction inside a loop counts once -- so they describe what a kernel is made of, not how long it
runs: how many matrix instructions stand against how many vector instructions in an MLP tile loop, how many independent global
loads the gather issues per sample, whether a kernel touches scratch, where DPP / LDS / atomics are used.

  python tools/instruction_mix.py [--out profiles/rNN_instruction_mix.txt] [--kernel substring]
"""
import argparse
import collections
import glob
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_resources import LLVM, ROOT, TARGET, demangle, short  # noqa: E402

CLASSES = ["mfma", "valu", "dpp", "salu", "waitcnt", "barrier", "vmem_ld", "vmem_st", "vmem_atomic", "lds", "lds_atomic", "scratch", "branch"]


def classify(mn, line):
    if mn in ("s_nop", "s_code_end", "s_endpgm"):  # padding behind a kernel / between its blocks: not part of the mix
        return None
    if mn.startswith("v_mfma") or mn.startswith("v_smfmac"):
        return "mfma"
    if mn.startswith("scratch_"):
        return "scratch"
    if mn.startswith(("global_atomic", "buffer_atomic", "flat_atomic")):
        return "vmem_atomic"
    if mn.startswith(("global_load", "buffer_load", "flat_load")):
        return "vmem_ld"
    if mn.startswith(("global_store", "buffer_store", "flat_store")):
        return "vmem_st"
    if mn.startswith("ds_"):
        return "lds_atomic" if re.match(r"ds_(add|sub|min|max|and|or|xor|pk_add|cmpst|inc|dec)", mn) else "lds"
    if mn == "s_waitcnt" or mn.startswith("s_wait"):
        return "waitcnt"
    if mn == "s_barrier":
        return "barrier"
    if mn.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if mn.startswith("s_"):
        return "salu"
    if mn.startswith("v_"):
        return "dpp" if ("_dpp" in mn or " row_" in line or "quad_perm" in line or "wave_sh" in line or "row_bcast" in line) else "valu"
    return None


def disassemble(obj_path):
    with tempfile.TemporaryDirectory() as tmp:
        fat, co = os.path.join(tmp, "fatbin"), os.path.join(tmp, "co")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", obj_path, fat], check=True)
        subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                        "--targets=" + TARGET, "--output=" + co], check=True)
        return subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--mcpu=gfx950", co], check=True, capture_output=True, text=True).stdout


def mix(variant=""):
    rows = []
    suffix = (".%s.o" % variant) if variant else ".o"
    for obj in sorted(glob.glob(os.path.join(ROOT, "f2-nerf_amd", "build", "*" + suffix))):
        base = os.path.basename(obj)
        if base.startswith("host_") or (not variant and base.count(".") != 1):
            continue
        cur, counts = None, None
        for line in disassemble(obj).splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
            if m:
                if cur and not cur.endswith(".kd"):
                    rows.append((base[:-len(suffix)] + ".hip", cur, counts))
                cur, counts = m.group(1), collections.Counter()
                continue
            if cur is None:
                continue
            m = re.match(r"^\s+([a-z_0-9]+)\s?(.*?)\s*//", line)
            if not m:
                continue
            c = classify(m.group(1), line)
            if c:
                counts[c] += 1
                counts["total"] += 1
        if cur and not cur.endswith(".kd"):
            rows.append((base[:-len(suffix)] + ".hip", cur, counts))
    names = demangle([r[1] for r in rows])
    return [(f, short(n), c) for (f, _, c), n in zip(rows, names)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variant", default="")
    ap.add_argument("--out", default="")
    ap.add_argument("--kernel", default="")
    a = ap.parse_args()
    rows = [r for r in mix(a.variant) if a.kernel in r[1]]
    head = "%-14s %-50s %6s " % ("file", "kernel", "total") + " ".join("%7s" % c[:7] for c in CLASSES)
    lines = ["# static instruction mix of the gfx950 kernels in libf2n_hip%s.so (tools/instruction_mix.py: llvm-objdump of the built code objects)" % (("_" + a.variant) if a.variant else ""),
             "# STATIC counts: an instruction in a loop counts once.  dpp = vector instructions with a DPP modifier (cross-lane inside a row / wave)",
             head]
    for f, n, c in sorted(rows):
        lines.append("%-14s %-50s %6d " % (f, n[:50], c["total"]) + " ".join("%7d" % c[k] for k in CLASSES))
    text = "\n".join(lines) + "\n"
    if a.out:
        with open(a.out, "w") as fh:
            fh.write(text)
    sys.stdout.write(text)


if __name__ == "__main__":
    main()
