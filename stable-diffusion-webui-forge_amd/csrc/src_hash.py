"""sha256 (first 16 hex digits) over csrc/*.hip, csrc/*.hpp and include/fmx.h: the identity of the kernel sources.  The Makefile bakes it into
libfmx_gfx950.so (fmx_build_info()), bench.py compares the loaded library's value with the one a committed PMC summary was taken on."""
import glob
import hashlib
import os

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def kernel_source_hash():
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(HERE, "*.hip")) + glob.glob(os.path.join(HERE, "*.hpp")) + [os.path.join(ROOT, "include", "fmx.h")]):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(kernel_source_hash())
