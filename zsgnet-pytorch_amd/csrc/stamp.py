"""Prints zsg_stamp.h: the sha256 stamp of the kernel sources (csrc/*.hip, *.h, *.cpp; basenames + contents, sorted) that libzsg.so is
being built from.  The Makefile compiles it into api.cpp (zsg_source_stamp()); ops.source_stamp() computes the same value from the files,
so a tuning table / rocprof summary can be tied to the LIBRARY that is loaded, not merely to the sources lying next to it."""
import glob
import hashlib
import os

here = os.path.dirname(os.path.abspath(__file__))
h = hashlib.sha256()
for f in sorted(glob.glob(os.path.join(here, "*"))):
    if f.endswith((".hip", ".h", ".cpp")):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
print(f'#define ZSG_SOURCE_STAMP "{h.hexdigest()[:16]}"')
