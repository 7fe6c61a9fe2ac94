#!/usr/bin/env python
"""Build libcoponerf_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python coponerf_amd/csrc/build.py [--force]

geometry.hip is compiled with -ffp-contract=off: its arithmetic must reproduce the oracle's
IEEE operation sequence bit for bit (sample coordinates / tap indices).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "libcoponerf_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
UNITS = [
    ("error.cpp", []),
    ("streams.cpp", []),
    ("geometry.hip", ["-ffp-contract=off"]),
    ("gather.hip", []),
    ("encode.hip", []),
    ("encode_fused.hip", []),
    ("encode_f32.hip", []),
    ("local_units.hip", []),
    ("encode_bwd.hip", []),
    ("gemm_f16.hip", []),
    ("attend.hip", []),
    ("linear_f32.hip", []),
    ("rayout.hip", []),
    ("ufc.hip", []),
    ("ufc_attn.hip", []),
    ("ufc_strided_bwd.hip", []),
    ("encoder.hip", []),
    ("input.hip", []),
    ("backward.hip", []),
    ("wgrad_tall.hip", []),
    ("wgrad_f32.hip", []),
    ("adam.hip", []),
    ("trunk_conv.hip", []),
    ("pose.hip", []),
]


def _newer(src, dst):
    return (not os.path.exists(dst)) or os.path.getmtime(src) > os.path.getmtime(dst)


def build(force=False, verbose=True):
    objs = []
    deps = [os.path.join(HERE, "common.h"), os.path.join(HERE, "taps.h"), os.path.join(HERE, "encode_common.h"), os.path.join(HERE, "..", "..", "include", "coponerf_hip.h")]
    for name, extra in UNITS:
        src = os.path.join(HERE, name)
        obj = os.path.join(HERE, os.path.splitext(name)[0] + ".o")
        if force or _newer(src, obj) or any(_newer(d, obj) for d in deps):
            cmd = [HIPCC, *COMMON, *extra, "-x", "hip", "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        objs.append(obj)
    if force or any(_newer(o, OUT) for o in objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", OUT]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
