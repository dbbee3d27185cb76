#!/usr/bin/env python3
"""Regenerate tests/golden/ref_primitives.json from the REFERENCE's own headers.

Runs only in the authoring container (needs /root/reference).  Builds oracle/_ref/ref_harness
(reference src/primitives.h + src/randGen.h + vendored FLANN, compiled where they lie) and
stores its output.  The fixture is data: inputs and the reference's outputs, as hex floats.
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if not os.path.isdir("/root/reference/src"):
    sys.exit("reference tree not present; the committed fixture stays as is")
subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
out = subprocess.check_output([os.path.join(ROOT, "oracle", "_ref", "ref_harness")])
with open(os.path.join(ROOT, "tests", "golden", "ref_primitives.json"), "wb") as f:
    f.write(out)
print("wrote ref_primitives.json", len(out), "bytes")
# node / tree API types (Node, Tree, Heap, DistanceHolder, SymmetricMatrix, Point(string)): oracle/types_harness.cpp
# built against the reference's own src/primitives.h + src/heap.h
out = subprocess.check_output([os.path.join(ROOT, "oracle", "_ref", "ref_types_harness")])
with open(os.path.join(ROOT, "tests", "golden", "ref_types.json"), "wb") as f:
    f.write(out)
print("wrote ref_types.json", len(out), "bytes")
