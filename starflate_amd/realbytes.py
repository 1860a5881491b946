"""Real bytes for the bench and the parity tests, built from files every copy of this image carries (there is no
network and no corpus in the container: BASELINE.json's alice29 / dickens / enwik / Silesia are absent).

  source : Python and C/C++ SOURCE TEXT -- the interpreter's standard library, torch's Python sources and the ROCm headers,
           concatenated in sorted path order behind a one-line header per file (a tar without the padding)
  binary : x86-64 machine code and read-only data -- the leading bytes of libtorch_cpu.so

Both are deterministic for a given image; `describe()` gives size and SHA-256 so two runs can be told to have seen the same
bytes.  The reference's integration test does the same thing with the one real file it has
(/root/reference/src/test/decompress_test.cpp:136-174: a real file in, bytes out, compared)."""
import hashlib
import os
import sysconfig

import numpy as np

_CACHE = {}


def _source_roots():
    roots = [(sysconfig.get_paths()["stdlib"], (".py",))]
    try:
        import torch

        roots.append((os.path.dirname(torch.__file__), (".py",)))
    except ImportError:
        pass
    roots.append(("/opt/rocm/include", (".h", ".hpp")))
    return roots


def source(limit=96 << 20):
    """Up to `limit` bytes of source text (uint8 array); shorter if the image holds less."""
    key = ("source", limit)
    if key in _CACHE:
        return _CACHE[key]
    parts, total = [], 0
    for root, exts in _source_roots():
        if not os.path.isdir(root):
            continue
        for r, dirs, files in os.walk(root):
            dirs[:] = sorted(d for d in dirs if d not in ("__pycache__", "site-packages", "dist-packages"))
            for name in sorted(files):
                if not name.endswith(exts):
                    continue
                p = os.path.join(r, name)
                try:
                    if os.path.islink(p):
                        continue
                    with open(p, "rb") as f:
                        body = f.read()
                except OSError:
                    continue
                head = f"==> {os.path.relpath(p, root)} ({len(body)} bytes) <==\n".encode()
                parts.append(head)
                parts.append(body)
                total += len(head) + len(body)
                if total >= limit:
                    break
            if total >= limit:
                break
        if total >= limit:
            break
    buf = np.frombuffer(b"".join(parts)[:limit], dtype=np.uint8).copy()
    _CACHE[key] = buf
    return buf


def binary(limit=256 << 20):
    """Up to `limit` bytes of x86-64 code + data (uint8 array): the head of libtorch_cpu.so; empty if it is not there."""
    key = ("binary", limit)
    if key in _CACHE:
        return _CACHE[key]
    buf = np.zeros(0, np.uint8)
    try:
        import torch

        p = os.path.join(os.path.dirname(torch.__file__), "lib", "libtorch_cpu.so")
        if os.path.isfile(p):
            buf = np.fromfile(p, dtype=np.uint8, count=limit)
    except ImportError:
        pass
    _CACHE[key] = buf
    return buf


def describe(buf, what):
    return f"{what}: {buf.size} bytes, sha256 {hashlib.sha256(buf.tobytes()).hexdigest()[:16]}"
