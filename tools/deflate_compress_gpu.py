#!/usr/bin/env python3
"""Drop-in for the reference's fixture generator (/root/reference/tools/deflate_compress.py: --src FILE [--fixed],
raw DEFLATE on stdout), producing the stream with the MI355X kernels instead of zlib.  Extra switches: --zlib /
--gzip put back the wrapper that tool strips; --block-bytes N sets the independently coded strip (sfh_options.block_bytes); --effort NAME the search effort
(sfh_options.effort: what zlib's level is to that tool's zlib.compress call);
--index FILE additionally saves the block index, the region sub-index and the strip size (numpy .npz) that let
`decompress()` run on the GPU."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(args):
    import numpy as np

    from starflate_amd import Compressor

    with open(args.src, "rb") as f:
        data = f.read()
    comp = Compressor(args.device)
    container = "zlib" if args.zlib else "gzip" if args.gzip else "raw"
    out = comp.compress(data, strategy="fixed" if args.fixed else "auto", container=container, block_bytes=args.block_bytes, effort=args.effort)
    if args.index:
        np.savez(args.index, offsets=comp.last_index(), regions=comp.last_subindex(), size=np.uint64(len(data)),
                 block_bytes=np.uint32(comp.last_block_bytes()))
    sys.stdout.buffer.write(out)


parser = argparse.ArgumentParser(prog="deflate_compress_gpu", description="generates deflate compressed data from a file (GPU)")
parser.add_argument("--src", help="path to input file", required=True)
parser.add_argument("--fixed", help="use fixed strategy", action="store_true")
parser.add_argument("--zlib", help="RFC 1950 wrapper", action="store_true")
parser.add_argument("--gzip", help="RFC 1952 wrapper", action="store_true")
parser.add_argument("--index", help="save block index + sub-index to this .npz")
parser.add_argument("--block-bytes", type=int, default=0, help="strip size, a multiple of 32768 (0: the library's default)")
parser.add_argument("--effort", default="default", choices=["default", "fast", "fastest", "thorough", "max", "best", "ultra", "extreme"],
                    help="search effort; best / ultra / extreme are exact hash chains of depth 8 / 16 / 32")
parser.add_argument("--device", type=int, default=0)

if __name__ == "__main__":
    main(parser.parse_args())
