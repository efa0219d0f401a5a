"""sha256 (first 16 hex digits) over the kernel sources whose timings / counters are stored under profiles/ and replayed into
bench.py's `roofline` object: a stored figure is printed only while these files are what they were when it was measured.
usage: python tools/src_hash.py        (prints the hash)"""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the product kernels of the Winograd layers (gemm_x3: split-bf16, conv_gemm: fp32 matrix cores), their callers, the stem + apply pass
FILES = ("gemm_x3.hip", "gemm_x3_bfrag.hip", "x3_tiles.h", "conv_gemm.hip", "conv_tiles.h", "winograd.hip", "group_norm.hip", "rn_common.h")


def kernel_source_hash():
    h = hashlib.sha256()
    for f in FILES:
        with open(os.path.join(ROOT, "retinanet-tensorflow_amd", "csrc", f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(kernel_source_hash())
