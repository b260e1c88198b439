#!/usr/bin/env python3
"""SHA-256 over every source the in-tree binaries are built from (sorted paths
+ contents). `make` stamps it into accessor-blas_amd/lib/SOURCE_HASH after a
build; tests/conftest.py rebuilds when the stamp does not match the tree, so a
green GPU run always corresponds to the kernels of this checkout."""
import hashlib
import os
import sys

REPO = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
ROOTS = ["accessor-blas_amd/csrc", "accessor-blas_amd/hip", "accessor-blas_amd/include",
         "accessor-blas_amd/tools/selftest_primitives.hip", "accessor-blas_amd/Makefile",
         "include"]


def source_hash():
    files = []
    for root in ROOTS:
        path = os.path.join(REPO, root)
        if os.path.isfile(path):
            files.append(path)
            continue
        for base, _, names in os.walk(path):
            files += [os.path.join(base, n) for n in names
                      if n.endswith((".hip", ".hpp", ".h", ".cpp"))]
    digest = hashlib.sha256()
    for path in sorted(files):
        digest.update(os.path.relpath(path, REPO).encode())
        with open(path, "rb") as f:
            digest.update(f.read())
    return digest.hexdigest()


if __name__ == "__main__":
    value = source_hash()
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as f:
            f.write(value + "\n")
    else:
        print(value)
