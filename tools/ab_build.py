#!/usr/bin/env python3
"""A/B builds of the library for kernel comparisons on ONE box: links waverange_amd/ab/ab_<tag>.so from the objects of the
last regular build, with one source file replaced.
usage: ab_build.py TAG FILE.hip|FILE.cpp [extra compiler flags...]   (FILE: a variant of the csrc file of the same basename
       after the first '@': e.g. /tmp/old@wr_fused.hip stands in for csrc/wr_fused.hip)
then:  WAVERANGE_AMD_LIB=waverange_amd/ab/ab_TAG.so python tools/prof_transform.py ..."""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from waverange_amd import build as B

tag, variant = sys.argv[1], sys.argv[2]
extra = sys.argv[3:]
name = os.path.basename(variant).split("@")[-1]
assert name in B.SOURCES, name
B.build(verbose=False)
bdir = os.path.join(B.HERE, "build")
obj = os.path.join(bdir, "ab_%s_%s.o" % (tag, name))
# the variant is compiled from inside csrc so that its relative includes resolve
tmp = os.path.join(B.CSRC, "_ab_%s_%s" % (tag, name))
with open(variant) as fh:
    src = fh.read()
with open(tmp, "w") as fh:
    fh.write(src)
try:
    subprocess.check_call([B._hipcc()] + B.COMMON + B.EXTRA.get(name, []) + extra + ["-c", tmp, "-o", obj])
finally:
    os.remove(tmp)
objs = [obj if s == name else os.path.join(bdir, s.replace("/", "_") + ".o") for s in B.SOURCES]
os.makedirs(os.path.join(B.HERE, "ab"), exist_ok=True)
out = os.path.join(B.HERE, "ab", "ab_%s.so" % tag)  # (waverange_amd/build/ does not travel with gpurun)
subprocess.check_call([B._hipcc(), "--offload-arch=gfx950", "-shared", "-o", out] + objs + ["-lhsa-runtime64", "-lpthread", "-Wl,-rpath,/opt/rocm/lib",
                                                                                          "-Wl,--version-script=" + os.path.join(B.CSRC, "exports.map")])
print(out)
