"""Compiles and runs the C++23 host-API tests (tests/cpp/*.cpp) with AMD clang -std=c++23:
the CPU one restates the reference's huffman/decompress unit tests, with and without
AddressSanitizer + UBSan (the reference's CI matrix, .github/workflows/check.yml:13-16);
the GPU one round-trips starflate::compress() through starflate::decompress()."""
import os
import shutil
import subprocess

import pytest

from conftest import GOLDEN, ROOT

CLANG = "/opt/rocm/llvm/bin/clang++"
FLAGS = ["-std=c++23", "-fno-exceptions", "-Wall", "-Wextra", "-Wpedantic", "-Wconversion", "-Werror",
         "-I" + os.path.join(ROOT, "include")]


def _have_clang():
    return os.path.exists(CLANG) or shutil.which("clang++")


@pytest.mark.parametrize("san", [[], ["-fsanitize=address,undefined", "-fno-sanitize-recover=all"]])
def test_host_api_cpu(tmp_path, san):
    if not _have_clang():
        pytest.skip("no clang++")
    exe = tmp_path / "host_api_test"
    subprocess.check_call([CLANG, "-O1", "-g"] + FLAGS + san + [os.path.join(ROOT, "tests", "cpp", "host_api_test.cpp"), "-o", str(exe)])
    # wrapped fixtures made by zlib / gzip themselves (the reference's tool strips these wrappers,
    # tools/deflate_compress.py:8-13); .named.gz carries FNAME so the optional-field skipping is exercised
    import gzip
    import io
    import zlib

    with open(os.path.join(GOLDEN, "starfleet.html"), "rb") as f:
        html = f.read()
    (tmp_path / "starfleet.html.zlib").write_bytes(zlib.compress(html, 6))
    (tmp_path / "starfleet.html.gz").write_bytes(gzip.compress(html, 6, mtime=0))
    buf = io.BytesIO()
    with gzip.GzipFile(filename="starfleet.html", mode="wb", fileobj=buf, mtime=1) as g:
        g.write(html)
    (tmp_path / "starfleet.html.named.gz").write_bytes(buf.getvalue())
    out = subprocess.run([str(exe), GOLDEN, str(tmp_path)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "0 failed" in out.stdout


@pytest.mark.gpu
def test_compress_roundtrip_cpp(tmp_path):
    from starflate_amd import build

    lib = build.build()
    exe = tmp_path / "compress_roundtrip"
    libdir = os.path.dirname(lib)
    # bind to the same HIP runtime torch ships, as the Python plumbing does, unless /opt/rocm is complete
    subprocess.check_call([CLANG, "-O2"] + FLAGS + [os.path.join(ROOT, "tests", "cpp", "compress_roundtrip.cpp"),
                                                   "-L" + libdir, "-lstarflate_hip", "-Wl,-rpath," + libdir, "-o", str(exe)])
    out = subprocess.run([str(exe), GOLDEN], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr


@pytest.mark.gpu
def test_gather_streams_cpp(tmp_path):
    """The C++23 multi-GPU binding (compress_device_async + gather_streams) over a 1-rank RCCL communicator."""
    import torch  # where the image keeps librccl.so / libamdhip64.so

    from starflate_amd import build

    lib = build.build()
    exe = tmp_path / "gather_streams"
    libdir, tlib = os.path.dirname(lib), os.path.join(os.path.dirname(torch.__file__), "lib")
    subprocess.check_call([CLANG, "-O2"] + FLAGS + [os.path.join(ROOT, "tests", "cpp", "gather_streams.cpp"),
                                                   "-L" + libdir, "-lstarflate_hip", "-ldl", "-Wl,-rpath," + libdir, "-o", str(exe)])
    env = dict(os.environ)
    if os.path.exists(os.path.join(tlib, "librccl.so")) and not os.path.exists("/opt/rocm/lib/librccl.so"):
        env["SFH_RCCL_LIB"] = os.path.join(tlib, "librccl.so")
    out = subprocess.run([str(exe), GOLDEN], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
