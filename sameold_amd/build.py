"""Builds the gfx950 shared library (sameold_amd/libsame_rx.so) in-tree with hipcc.

    python -m sameold_amd.build [--force]

-ffp-contract=off is part of the arithmetic contract (host and device): the kernels
reproduce the reference's one-rounding-per-operation f32 arithmetic bit for bit.
"""
from __future__ import annotations

import hashlib
import os
import re
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# SAME_BUILD_VARIANT=name (measurement builds: SAME_PROFILE / SAME_SYM_TL ...) keeps its objects and library apart from the
# product's -- build/name/, libsame_rx.name.so -- and tools load it with SAME_LIB_VARIANT=name; the product never reads either.
VARIANT = os.environ.get("SAME_BUILD_VARIANT", "")
LIB = os.path.join(HERE, f"libsame_rx.{VARIANT}.so" if VARIANT else "libsame_rx.so")
SOURCES = ["same_kernels.hip", "same_kernels_fast.hip", "same_kernels_pipe.hip", "same_kernels_relaxed.hip", "same_kernels_sym.hip", "same_kernels_sym_hi.hip", "same_synth.hip", "same_batch.cpp", "same_config.cpp", "same_transport.cpp", "same_place.cpp"]
# flags of one source only.  same_kernels_sym_hi.hip (the symbol-paced pipeline at 44.1 / 48 kHz, six wavefronts per CU): the
# machine scheduler set for instruction-level parallelism -- 3.3-3.6 % faster there with the same events; the 22.05 kHz unit
# (twelve wavefronts per CU) measured up to 10 % SLOWER with it and keeps the default (same_kernels_sym.hip, SYM_SPLIT_TU)
# same_kernels_pipe.hip (the strict wavefront pipeline: a few wavefronts per CU, every one waiting on its own dependent chains):
# the same option, 0-2 % faster (configs[2] strict 5.79 -> 5.67 ms, the 32 768-channel shard 4.05 -> 4.02, configs[1] equal); the
# operations and their order are the source's either way (-ffp-contract=off: the scheduler reorders, it does not reassociate)
SOURCE_FLAGS = {"same_kernels_sym_hi.hip": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"],
                "same_kernels_pipe.hip": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]}
HEADERS = ["same_dev_common.h", "same_fast_common.h", "same_relaxed_common.h", "same_pipe_common.h", "same_profile.h", "same_device.h", "same_launch.h", "same_config.h", "same_transport.h", "../../include/same_rx.h",
           "../../include/same_place.h", "samedec_main.cpp"]
SAMEDEC = os.path.join(HERE, "samedec_gpu")      # the command-line decoder (host-only program, dlopens LIB)
ARCH = "gfx950"


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the MI355X library cannot be built (there is no CPU build)")


def flags() -> list:
    return [
        f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC",
        "-ffp-contract=off", "-fno-fast-math",
        "-fhip-fp32-correctly-rounded-divide-sqrt",
        "-fgpu-rdc" if False else "-fno-gpu-rdc",
        "-Wall", "-Wno-unused-function", "-Wno-unused-variable",
        "-x", "hip",
    ] + (["-DSAME_PROFILE=1"] if os.environ.get("SAME_PROFILE") else []) \
      + (["-DSAME_P3_MARKS=1"] if os.environ.get("SAME_P3_MARKS") else []) \
      + (["-DSAME_P1_SPLIT=1"] if os.environ.get("SAME_P1_SPLIT") else []) \
      + (["-DSAME_SYM_TL=1"] if os.environ.get("SAME_SYM_TL") else []) \
      + ([f"-DSYM_PRIOS=0x{os.environ['SAME_SYM_PRIOS']}"] if os.environ.get("SAME_SYM_PRIOS") else []) \
      + ([f"-DSYM_TL_GROUP={int(os.environ['SAME_SYM_TL_GROUP'])}u"] if os.environ.get("SAME_SYM_TL_GROUP") else []) \
      + [f"-D{d}" for d in os.environ.get("SAME_EXTRA_DEFS", "").split() if d] \
      + [f for f in os.environ.get("SAME_EXTRA_FLAGS", "").split() if f]      # (A/B builds of a variant: SAME_BUILD_VARIANT; raw compiler flags, e.g. "-mllvm -amdgpu-sched-strategy=max-ilp")


def source_hash() -> str:
    """sha256 over everything the library is made of: sources, headers, this script's flags."""
    h = hashlib.sha256()
    for f in sorted(SOURCES + HEADERS):
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read() + b"\0")
    h.update(" ".join(flags()).encode())
    h.update(repr(sorted(SOURCE_FLAGS.items())).encode())
    return h.hexdigest()


_MARK = re.compile(rb"SAME_SOURCE_HASH=([0-9a-f]{64})")


def built_hash(path: str = LIB):
    """The source hash compiled into an existing library (same_rx_source_hash()), read from the file
    itself so that nothing has to be loaded; None when the file is missing or predates the scheme."""
    try:
        with open(path, "rb") as fh:
            m = _MARK.search(fh.read())
    except OSError:
        return None
    return m.group(1).decode() if m else None


def is_stale() -> bool:
    """The shipped binary must be the one HEAD's sources build: compare hashes, not mtimes (the .so
    travels to the GPU box with fresh mtimes and is git-ignored)."""
    if not os.path.exists(LIB) or not os.path.exists(SAMEDEC):
        return True
    return built_hash() != source_hash()


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not is_stale():
        return LIB
    objs = []
    digest = source_hash()
    cc = hipcc()
    bdir = os.path.join(HERE, "build", VARIANT) if VARIANT else os.path.join(HERE, "build")
    os.makedirs(bdir, exist_ok=True)
    procs = []
    # Objects are re-used when nothing they are made of changed: the key is the source, every header (any of them may be
    # included) and the flags; only same_batch.cpp carries the library-wide source hash (same_rx_source_hash()).
    hdr = hashlib.sha256()
    for f in sorted(HEADERS + ["same_kernels_sym.hip"]):      # (same_kernels_sym_hi.hip includes it)
        with open(os.path.join(CSRC, f), "rb") as fh:
            hdr.update(f.encode() + b"\0" + fh.read() + b"\0")
    hdr.update(" ".join(flags()).encode())
    for src in SOURCES:
        obj = os.path.join(bdir, os.path.splitext(src)[0] + ".o")
        objs.append(obj)
        extra = ([f'-DSAME_SOURCE_HASH="{digest}"'] if src == "same_batch.cpp" else []) + SOURCE_FLAGS.get(src, [])
        with open(os.path.join(CSRC, src), "rb") as fh:
            key = hashlib.sha256(hdr.digest() + fh.read() + " ".join(extra).encode()).hexdigest()
        stamp = obj + ".key"
        if not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == key:
            continue
        cmd = [cc] + flags() + extra + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, stamp, key, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = None
    for src, stamp, key, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode(errors="replace"))
            failed = failed or src
            continue
        with open(stamp, "w") as fh:
            fh.write(key)
        if verbose and out:
            print(out.decode(errors="replace"))
    if failed:
        raise RuntimeError(f"hipcc failed on {failed}")
    cmd = [cc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs
    subprocess.run(cmd, check=True)
    cxx = shutil.which("g++") or cc
    subprocess.run([cxx, "-O2", "-std=c++17", "-Wall", os.path.join(CSRC, "samedec_main.cpp"), "-o", SAMEDEC, "-ldl"], check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
