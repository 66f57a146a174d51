#!/usr/bin/env python3
"""Builds the CPU checkers (test infrastructure, never loaded by the product path):
oracle/_tclip_oracle.so (C++ restatement of the loop) and oracle/_mathcheck.so (host build of the
product's special-function header for value-by-value comparison with torch)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
FLAGS = ["-O2", "-ffp-contract=off", "-mfma", "-fopenmp", "-shared", "-fPIC"]
CSRC = os.path.join(HERE, "..", "transductive-clip_amd", "csrc")
HEADERS = [os.path.join(CSRC, h) for h in ("tclip_math.h", "tclip_selftest_inputs.h", "tclip_rsqrt14_table.h", "tclip_rsqrt14_table_dev.h",
                                             "tclip_rcp14_log_table.h")]


def _build(src, out):
    src, out = os.path.join(HERE, src), os.path.join(HERE, out)
    if os.path.exists(out) and all(os.path.getmtime(d) <= os.path.getmtime(out) for d in [src] + HEADERS):
        return out
    subprocess.check_call(["g++"] + FLAGS + ["-o", out, src, "-lm"])
    return out


def build():
    return _build("tclip_oracle.cpp", "_tclip_oracle.so"), _build("mathcheck.cpp", "_mathcheck.so")


if __name__ == "__main__":
    print(build())
