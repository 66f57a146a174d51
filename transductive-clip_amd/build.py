#!/usr/bin/env python3
"""Builds the HIP library of the engine in-tree: tclip_amd/libtclip.so (gfx950 code object + C ABI).

hipcc cross-compiles without a GPU.  -ffp-contract=off is part of the numerics contract: the
kernels reproduce the reference's separately rounded fp32 operations and spell every fused
multiply-add they want explicitly (csrc/tclip_math.h).  -fno-slp-vectorize: the kernels spell the packed fp32 operations they
want themselves (csrc/tclip_pk.h); what the SLP vectoriser packed on its own in the MM kernels cost more in register moves and
explicit negations than the packed instruction saved (gfx950 issues one v_pk_*_f32 in the time of two scalar ones), and its
packed double-float code spilled (HISTORY.md section 5, round 4)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "tclip_amd", "libtclip.so")
SOURCES = ["tclip_kernels.hip", "tclip_host.cpp"]
HEADERS = ["tclip_math.h", "tclip_device.h", "tclip_pk.h", "tclip_tim.inc", "tclip_lshot.inc", "tclip_selftest_inputs.h", "tclip_rsqrt14_table.h", "tclip_rsqrt14_table_dev.h", "tclip_rcp14_log_table.h", os.path.join("..", "..", "include", "tclip.h")]


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def up_to_date():
    if not os.path.exists(OUT):
        return False
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return all(os.path.getmtime(d) <= t for d in deps)


def build(force=False, verbose=False):
    if not force and up_to_date():
        return OUT
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-shared", "-pthread", "-std=c++17",
           "-Wall", "-Wno-unused-function", "-o", OUT] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd, cwd=CSRC)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
