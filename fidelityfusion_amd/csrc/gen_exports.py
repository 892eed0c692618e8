#!/usr/bin/env python3
"""Linker version script for libffgp.so: exactly the functions include/ffgp.h declares are exported (everything else -- internal
C++ functions, HIP kernel handle objects and device stubs, template instantiations of the standard library -- stays local)."""
import re
import sys

src = re.sub(r"/\*.*?\*/", "", open(sys.argv[1]).read(), flags=re.S)
names = sorted(set(re.findall(r"\b(ffgp_[a-z0-9_]+)\s*\(", src)))
print("{ global:\n" + "".join("    %s;\n" % n for n in names) + "  local: *;\n};")
