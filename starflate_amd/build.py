"""Builds libstarflate_hip.so in-tree with hipcc for gfx950 (no JIT cache, no cmake)."""
import os
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
# SFH_LIB=<path>: another build of the library (kernel experiments side by side: tools/exp/variants.sh)
LIB_PATH = os.environ.get("SFH_LIB") or os.path.join(PKG_DIR, "libstarflate_hip.so")
SOURCES = ["sf_kernels.hip", "sf_checksum.hip", "sf_inflate.hip", "sf_capi.hip"]
HEADERS = [os.path.join(CSRC, "sf_device.h"), os.path.join(CSRC, "sf_inflate_core.h"),
           os.path.join(os.path.dirname(PKG_DIR), "include", "starflate_hip.h")]


def hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """Compile every HIP source for gfx950 into starflate_amd/libstarflate_hip.so."""
    if not force and not needs_build():
        return LIB_PATH
    cmd = [hipcc(), "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared",
           "-Wall", "-Wextra", "-Werror"]
    cmd += os.environ.get("SF_HIPCC_FLAGS", "").split()
    cmd += [os.path.join(CSRC, s) for s in SOURCES]
    cmd += ["-o", LIB_PATH + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
