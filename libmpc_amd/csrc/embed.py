"""Turns the engine headers into a C++ initialiser list of (name, text) pairs: what nlmpc_jit.cpp hands to hipRTC as
in-memory headers when it compiles user-supplied hook sources at run time."""
import os
import sys

out, files = sys.argv[1], sys.argv[2:]
with open(out, "w") as f:
    for path in files:
        name = "mpcx/" + os.path.basename(path)
        data = open(path, "rb").read()
        f.write('{"%s", {' % name)
        f.write(",".join(str(b) for b in data))
        f.write(",0}},\n")
