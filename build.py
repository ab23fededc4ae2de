#!/usr/bin/env python3
"""Build libdiffsci_hip.so (gfx950) in-tree with hipcc, and the oracle side-builds.

    python build.py            # compile if sources are newer than the library
    python build.py --force
The shared library lands in diffsci_amd/_lib/ (git-ignored; travels to the GPU box with
the gpurun snapshot).
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(ROOT, "diffsci_amd", "csrc")
OUTDIR = os.path.join(ROOT, "diffsci_amd", "_lib")
LIB = os.path.join(OUTDIR, "libdiffsci_hip.so")
SOURCES = ["ds_api.hip", "ds_step.hip", "ds_norm.hip", "ds_gnorm.hip", "ds_normtab.hip", "ds_conv.hip", "ds_conv6.hip", "ds_conv3h.hip", "ds_conv3p.hip", "ds_convup.hip", "ds_conv1h.hip", "ds_convdirect.hip", "ds_conv3d.hip", "ds_attn.hip", "ds_attn3h.hip", "ds_small.hip", "ds_amax.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -ffp-contract=off: the stepper / norm kernels reproduce the reference's one-rounding-per-op
# arithmetic; MFMA kernels are unaffected (their FMAs are the matrix instruction's own).
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-std=c++17",
         "-Wall", "-Wno-unused-function"] + os.environ.get("DS_EXTRA_HIPCC_FLAGS", "").split()     # measurement builds (-DDS_...=1)
# The fp16x3 convolution kernels: no SLP vectorisation.  It packs the epilogue's pairs of 16-lane DPP reductions into v_pk_add_f32,
# which cannot carry the DPP modifier (two v_mov_b32_dpp + one packed add per step instead of two v_add_f32_dpp), and packed fp32
# operations issue no faster than their two scalar halves on gfx950.  Same-box A/B: 77.4 -> 78.5 samples/s, launches 223.4 -> 219.3 us
# (profiles/r02_noslp_ab.log); the attention kernel measured neutral with the flag and keeps the default.
EXTRA_FLAGS = {f: ["-fno-slp-vectorize"] for f in ("ds_conv3h.hip", "ds_conv3p.hip", "ds_convup.hip", "ds_conv1h.hip")}
# ds_conv3p.hip issues its LDS-DMA from inline assembly (the reason is in the file) and names m0, which that instruction reads, as
# clobbered; clang warns that m0 is a reserved register.  No compiler-generated code of that kernel uses m0.
EXTRA_FLAGS["ds_conv3p.hip"] = EXTRA_FLAGS["ds_conv3p.hip"] + ["-Wno-inline-asm"]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(ROOT, "include", "diffsci_hip.h"), os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    os.makedirs(OUTDIR, exist_ok=True)
    if not force and not _stale():
        return LIB
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(OUTDIR, src.replace(".hip", ".o"))
        cmd = [HIPCC] + FLAGS + EXTRA_FLAGS.get(src, []) + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd)))
        objs.append(obj)
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
