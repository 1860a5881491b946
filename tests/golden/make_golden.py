#!/usr/bin/env python3
"""Regenerates the golden fixtures from the reference checkout (build container only;
/root/reference does not exist on the GPU box, the committed outputs travel instead).

  starfleet.html          data file of the reference's own test
                          (/root/reference/src/test/starfleet.html, used by
                          src/test/decompress_test.cpp:136-174)
  starfleet.html.dynamic  output of the reference's fixture tool
  starfleet.html.fixed      /root/reference/tools/deflate_compress.py [--fixed]
                          (what tools/compressed_file.bzl:28-36 runs at build time)
"""
import os
import shutil
import subprocess
import sys

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


if __name__ == "__main__":
    main()
