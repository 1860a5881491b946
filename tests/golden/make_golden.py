#!/usr/bin/env python3
"""Regenerates the golden fixtures from the reference checkout (build container only;
/root/reference does not exist on the GPU box, the committed outputs travel instead).

  starfleet.html          data file of the reference's own test
                          (/root/reference/src/test/starfleet.html, used by
                          src/test/decompress_test.cpp:136-174)
  starfleet.html.dynamic  output of the reference's fixture tool
  starfleet.html.fixed      /root/reference/tools/deflate_compress.py [--fixed]
                          (what tools/compressed_file.bzl:28-36 runs at build time)
  starfleet.html.dynamic.flushed / .fixed.flushed (+ .index: little-endian u64 stream offsets)
                          the same bytes through the same zlib settings as that tool
                          (tools/deflate_compress.py:8-13: wbits=-MAX_WBITS, default level, strategy
                          default / Z_FIXED), with Z_FULL_FLUSH after every 32 KiB of input: the stream the
                          reference's test decodes (src/test/decompress_test.cpp:136-174) in a form whose
                          32 KiB segments are independently decodable -- what the GPU decoder reads.  The tool
                          itself has no flush option, so its compressobj call is restated here; the checks at
                          the bottom tie the two together (same data, same settings, inflate to the same bytes).
"""
import os
import shutil
import struct
import subprocess
import sys
import zlib

REF = os.environ.get("STARFLATE_REFERENCE_DIR", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    src = os.path.join(REF, "src", "test", "starfleet.html")
    tool = os.path.join(REF, "tools", "deflate_compress.py")
    shutil.copyfile(src, os.path.join(HERE, "starfleet.html"))
    for name, extra in (("starfleet.html.dynamic", []), ("starfleet.html.fixed", ["--fixed"])):
        out = subprocess.check_output([sys.executable, tool, "--src", src] + extra)
        with open(os.path.join(HERE, name), "wb") as f:
            f.write(out)
        print(name, len(out))
    with open(src, "rb") as f:
        data = f.read()
    seg = 32768
    for name, strategy in (("starfleet.html.dynamic", zlib.Z_DEFAULT_STRATEGY), ("starfleet.html.fixed", zlib.Z_FIXED)):
        co = zlib.compressobj(wbits=-zlib.MAX_WBITS, strategy=strategy)  # the tool's call, tools/deflate_compress.py:8-13
        parts = []
        for at in range(0, len(data), seg):
            last = at + seg >= len(data)
            parts.append(co.compress(data[at:at + seg]) + co.flush(zlib.Z_FINISH if last else zlib.Z_FULL_FLUSH))
        offsets = [0]
        for p in parts:
            offsets.append(offsets[-1] + len(p))
        stream = b"".join(parts)
        with open(os.path.join(HERE, name), "rb") as f:  # what the tool itself wrote: same bytes once inflated
            assert zlib.decompress(f.read(), -15) == zlib.decompress(stream, -15) == data
        with open(os.path.join(HERE, name + ".flushed"), "wb") as f:
            f.write(stream)
        with open(os.path.join(HERE, name + ".flushed.index"), "wb") as f:
            f.write(struct.pack(f"<{len(offsets)}Q", *offsets))
        print(name + ".flushed", len(stream), "segments", len(parts))


if __name__ == "__main__":
    main()
