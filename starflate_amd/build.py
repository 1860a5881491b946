"""Builds libstarflate_hip.so in-tree with hipcc for gfx950 (no JIT cache, no cmake)."""
import os
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
# SFH_LIB=<path>: another build of the library (kernel experiments side by side: tools/exp/variants.sh)
LIB_PATH = os.environ.get("SFH_LIB") or os.path.join(PKG_DIR, "libstarflate_hip.so")
SOURCES = ["sf_kernels.hip", "sf_checksum.hip", "sf_inflate.hip", "sf_guard.hip", "sf_capi.hip"]
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", "-Wall", "-Wextra", "-Werror"]
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
    cmd = [hipcc()] + FLAGS
    cmd += os.environ.get("SF_HIPCC_FLAGS", "").split()
    cmd += [os.path.join(CSRC, s) for s in SOURCES]
    cmd += ["-o", LIB_PATH + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    return LIB_PATH


def source_stamp():
    """What ties a measurement to the code it measured: the commit (git here; on a GPU box, which has no .git, the
    `.commit_stamp` file the post-commit hook / tools/stamp_commit.sh leaves at the repo root) and a SHA-256 over the
    kernel sources, the C-ABI header and the compiler flags of build() (SF_HIPCC_FLAGS of a variant build included) -- the same
    on both sides whatever the commit is called.  (The ROCm version is the image's, the same here and on the GPU box.)"""
    import hashlib

    root = os.path.dirname(PKG_DIR)
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h")))
    files.append(os.path.join(root, "include", "starflate_hip.h"))
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(FLAGS + os.environ.get("SF_HIPCC_FLAGS", "").split()).encode())
    commit, dirty = None, None
    try:
        commit = subprocess.check_output(["git", "-C", root, "rev-parse", "--short", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
        dirty = bool(subprocess.check_output(["git", "-C", root, "status", "--porcelain", "--", "starflate_amd/csrc", "include"],
                                             stderr=subprocess.DEVNULL).decode().strip())
    except Exception:  # noqa: BLE001  (no .git on the GPU box)
        try:
            with open(os.path.join(root, ".commit_stamp")) as fh:
                commit = fh.read().strip() or None
        except OSError:
            pass
    return {"commit": commit or "unknown", "dirty": dirty, "csrc_sha256": h.hexdigest()[:16]}


if __name__ == "__main__":
    print(build(force=True, verbose=True))
