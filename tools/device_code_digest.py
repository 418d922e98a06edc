"""SHA-256 of the gfx950 machine code (.text of the code object) of every kernel source of the product library, as hipcc has just built
it.  Two builds with equal digests run the same kernels: what a record such as profiles/r05_device_code_digest.txt is for -- commits that
follow a measured one and touch only host code can be checked, without a GPU, to have left every kernel as it was measured and tested.

  python tools/device_code_digest.py [--variant ''|refnum|debug] [--check profiles/rNN_device_code_digest.txt]
"""
import argparse
import glob
import hashlib
import os
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_resources import LLVM, ROOT, TARGET  # noqa: E402


def digest(obj_path):
    with tempfile.TemporaryDirectory() as tmp:
        fat, co, text = (os.path.join(tmp, n) for n in ("fatbin", "co", "text"))
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", obj_path, fat], check=True)
        subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                        "--targets=" + TARGET, "--output=" + co], check=True)
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.text", co, text], check=True)
        data = open(text, "rb").read()
    return hashlib.sha256(data).hexdigest(), len(data)


def table(variant=""):
    suffix = (".%s.o" % variant) if variant else ".o"
    out = {}
    for obj in sorted(glob.glob(os.path.join(ROOT, "f2-nerf_amd", "build", "*" + suffix))):
        base = os.path.basename(obj)
        if base.startswith("host_") or (not variant and base.count(".") != 1):
            continue
        out[base[:-len(suffix)] + ".hip"] = digest(obj)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variant", default="")
    ap.add_argument("--check", default="")
    a = ap.parse_args()
    t = table(a.variant)
    for f, (h, n) in t.items():
        print("%-18s %s %8d bytes of gfx950 code" % (f, h, n))
    if a.check:
        want = {l.split()[0]: l.split()[1] for l in open(a.check) if l.strip() and not l.startswith("#")}
        bad = [f for f in t if want.get(f) != t[f][0]]
        print("differs from %s: %s" % (a.check, ", ".join(bad) if bad else "nothing"))
        sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
