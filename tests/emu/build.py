"""tests/emu/build.py -- TEST INFRASTRUCTURE ONLY: builds the HOST TWIN of the demodulator (tests/emu/_build/libsdhip_emu.so)
from the unchanged HIP sources satdump_amd/csrc/demod_{kernels,engine}.hip and the stand-in runtime in this directory.
The only textual change: the two `asm volatile("" : "+s"(z))` scheduling barriers (an AMDGPU register constraint) are dropped."""
import os
import re
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
# EMU_CSRC / EMU_TAG: build a twin of ANOTHER copy of the sources (e.g. an older commit, for A/B on the CPU) next to the default one
CSRC = os.environ.get("EMU_CSRC") or os.path.join(ROOT, "satdump_amd", "csrc")
OUT = os.path.join(HERE, "_build" + os.environ.get("EMU_TAG", ""))
LIB = os.path.join(OUT, "libsdhip_emu.so")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
SOURCES = ["demod_kernels.hip", "demod_engine.hip", "fec_kernels.hip", "fec_engine.hip", "dvbs2_ldpc.hip", "dvbs2_bch.hip", "dvbs2_demap.hip", "dvbs2_engine.hip", "dvbs2_ts.hip", "shard.hip", "aos_demux.hip", "lrpt_decoder.hip"]


def build(force: bool = False) -> str:
    os.makedirs(OUT, exist_ok=True)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "emu_runtime.cpp"), os.path.join(HERE, "hip", "hip_runtime.h"),
                                                                  os.path.join(ROOT, "include", "sdhip.h"), os.path.abspath(__file__)]
    fresh = lambda: os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(d) for d in deps)  # noqa: E731
    if not force and fresh():
        return LIB
    # several test processes (pytest-xdist workers) may arrive here at once: one builds, the others wait for it; the library appears atomically
    import fcntl
    with open(os.path.join(OUT, ".build.lock"), "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        if not force and fresh():
            return LIB
        return _build_locked()


def _build_locked() -> str:
    gen = []
    for f in SOURCES:
        src = open(os.path.join(CSRC, f)).read()
        src = re.sub(r'asm volatile\("" : "\+s"\(\w+\)\);', "", src)
        src = re.sub(r'__attribute__\(\(amdgpu_waves_per_eu\([^)]*\)\)\)', "", src)
        # constants pinned into registers with v_mov / s_mov: plain assignments here
        src = re.sub(r'asm volatile\("[sv]_mov_b32 %0, (0x[0-9a-fA-F]+)" : "=[sv]"\(([^;]+?)\)\);', r"\2 = \1;", src)
        # a scalar copied into a vector register (sd_to_vgpr): a plain copy here
        src = re.sub(r'asm\("v_mov_b32 %0, %1" : "=v"\((\w+)\) : "s"\((\w+)\)\);', r"\1 = \2;", src)
        dst = os.path.join(OUT, f.replace(".hip", "_emu.cpp"))
        open(dst, "w").write(src)
        gen.append(dst)
    cmd = [CLANG, "-x", "c++", "-std=c++17", "-O1", "-g", "-ffp-contract=off", "-fPIC", "-shared", "-Wno-unknown-attributes", "-D__HIPCC__", "-DSDHIP_HOST_TWIN", "-Wno-unused-value",
           "-I", HERE, "-I", CSRC, "-I", os.path.join(ROOT, "include"), "-o", LIB + ".tmp"] + os.environ.get("EMU_DEFS", "").split() + gen + [os.path.join(HERE, "emu_runtime.cpp"), "-lm"]
    subprocess.run(cmd, check=True)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force=True))
