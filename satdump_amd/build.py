"""Build recipe for the native pieces (run by __graft_entry__.build(); hipcc cross-compiles gfx950 without a GPU).

  satdump_amd/lib/libsdhip.so   the product: HIP kernels + host engines + C ABI (include/sdhip.h)
  oracle/_build/libsdoracle.so  test oracle: plain-C restatement (building the checker is not using it)
  oracle/_ref/libsdref.so       test oracle: the reference's own sources, only when /root/reference exists
"""
from __future__ import annotations

import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "satdump_amd", "csrc")
LIBDIR = os.path.join(ROOT, "satdump_amd", "lib")
LIB = os.path.join(LIBDIR, "libsdhip.so")

HIP_SOURCES = ["fec_kernels.hip", "fec_engine.hip", "demod_kernels.hip", "demod_engine.hip", "dvbs2_ldpc.hip", "dvbs2_bch.hip", "dvbs2_demap.hip", "dvbs2_engine.hip", "dvbs2_ts.hip", "shard.hip", "aos_demux.hip", "lrpt_decoder.hip"]
# -ffp-contract=off: the float chains must round exactly where the reference's x86-64 -O2 build rounds (no FMA)
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-Wno-unused-result"]


def _newer(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _compile_one(args):
    hipcc, src, obj, verbose = args
    cmd = [hipcc] + [f for f in HIPCC_FLAGS if f != "-shared"] + ["-c", src, "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd, cwd=ROOT)
    return obj


def build_lib(force: bool = False, verbose: bool = True) -> str:
    """One object per translation unit (compiled side by side, only the ones whose source or a header changed), then one link."""
    from concurrent.futures import ThreadPoolExecutor
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    srcs = [os.path.join(CSRC, s) for s in HIP_SOURCES if os.path.exists(os.path.join(CSRC, s))]
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))] + [os.path.join(ROOT, "include", "sdhip.h")]
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    jobs, objs = [], []
    for s in srcs:
        o = os.path.join(objdir, os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        if force or _newer(o, [s] + hdrs):
            jobs.append((hipcc, s, o, verbose))
    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
            list(ex.map(_compile_one, jobs))
    if jobs or force or _newer(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-fPIC", "-shared"] + objs + ["-o", LIB]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd, cwd=ROOT)
    return LIB


def source_hash() -> str:
    """Fingerprint of the kernel sources libsdhip.so is built from (csrc/*.hip, csrc/*.h, include/sdhip.h, the hipcc flags):
    tools/pmc_summary.py stamps it into profiles/*_pmc.csv and bench.py only quotes a PMC traffic figure whose stamp matches."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))) 
    for f in files:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    with open(os.path.join(ROOT, "include", "sdhip.h"), "rb") as fh:
        h.update(fh.read())
    h.update(" ".join(HIPCC_FLAGS).encode())
    return h.hexdigest()[:16]


def build_oracle(verbose: bool = True) -> None:
    cmd = ["make", "-C", os.path.join(ROOT, "oracle"), "-j8", "all"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def build_plugin_check(verbose: bool = True) -> None:
    """Compile the SatDump plugin shim (plugin/sdhip_plugin.cpp) against the reference's own headers: a build check of the
    drop-in boundary, only possible where the reference tree exists (not on the GPU box)."""
    if not os.path.isdir("/root/reference/src-core"):
        return
    for d in (os.path.join(ROOT, "plugin"), os.path.join(ROOT, "tests", "minihost")):  # the plugin, and the test host that runs it
        cmd = ["make", "-C", d, "all"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)


def build_all(force: bool = False) -> None:
    build_lib(force=force)
    build_oracle()
    build_plugin_check()


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
