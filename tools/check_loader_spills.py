#!/usr/bin/env python3
"""Build-time check of the collapsed kernel's loader wave.

Its source loads are hand-written instructions whose completion the compiler
does not track: a register that such a load is still filling must never be
spilled (the spill would store whatever the register held before).  Compiles
the fast kernels to assembly and fails if any instance of
drb_collapsed_plain_kernel stores to scratch memory after the point where the
loader wave's code begins (its `s_setprio 3`)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "audio-resampler_amd", "csrc")


def main():
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "fast.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17",
                               "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "--cuda-device-only", "-S",
                               "-o", out, os.path.join(CSRC, "drb_kernels_fast.hip")], stderr=subprocess.DEVNULL)
        lines = open(out).read().split("\n")
    name, start, bad, seen = None, 0, [], 0
    for i, line in enumerate(lines):
        match = re.match(r"^(_ZN12_GLOBAL__N_126drb_collapsed_plain_kernel\w+):", line)
        if match:
            name, start = match.group(1), i
        if name and ".end_amdhsa_kernel" in line:
            body = lines[start:i]
            loader = [k for k, text in enumerate(body) if "s_setprio 3" in text]
            assert loader, "no loader wave found in " + name
            stores = [text for text in body[loader[0]:] if "scratch_store" in text]
            if stores:
                bad.append((name, len(stores)))
            seen += 1
            name = None
    assert seen > 0, "no drb_collapsed_plain_kernel instance found"
    for kernel, count in bad:
        print("SPILL in the loader of", kernel, ":", count, "scratch stores")
    print("%d kernel instances checked, %d with spills in the loader" % (seen, len(bad)))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
