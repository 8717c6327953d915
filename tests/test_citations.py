"""Every `file:line` citation of the reference (headers, docs, docstrings) must point into an existing file of the
reference tree and inside its length.  Runs only where /root/reference is mounted (the build container); the GPU box
does not have it and nothing else in the test suite or the product reads it."""
import glob
import os
import re

import pytest

from conftest import ROOT

REF = "/root/reference"
PAT = re.compile(r"((?:src|include/DPGO|examples|tests|cmake)/[A-Za-z0-9_/]+\.(?:cpp|h|cmake)|CMakeLists\.txt):(\d+)(?:-(\d+))?")


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not mounted")
def test_reference_citations_exist():
    files = [os.path.join(ROOT, f) for f in ("DESIGN.md", "INTEGRATION.md", "README.md", "bench.py", "__graft_entry__.py")]
    for pat in ("include/*.h", "include/*.hpp", "dpgo_amd/*.py", "dpgo_amd/csrc/*.hip", "dpgo_amd/csrc/*.cpp",
                "dpgo_amd/csrc/*.h", "dpgo_amd/csrc/kernels/*.h", "oracle/*.py", "oracle/*.c", "tests/*.py", "tests/cxx/*.cpp",
                "examples/*.py"):
        files += glob.glob(os.path.join(ROOT, pat))
    lengths, bad, n = {}, [], 0
    for f in files:
        if os.path.basename(f) == "test_citations.py":
            continue
        text = open(f, errors="replace").read()
        for m in PAT.finditer(text):
            rel, lo, hi = m.group(1), int(m.group(2)), int(m.group(3) or m.group(2))
            # the repo's own tests/ and include/ paths are not reference citations
            if rel.startswith("tests/") and os.path.exists(os.path.join(ROOT, rel)):
                continue
            path = os.path.join(REF, rel)
            n += 1
            if not os.path.isfile(path):
                bad.append("%s: %s does not exist in the reference" % (os.path.relpath(f, ROOT), m.group(0)))
                continue
            if path not in lengths:
                lengths[path] = sum(1 for _ in open(path, errors="replace"))
            if lo < 1 or hi < lo or hi > lengths[path]:
                bad.append("%s: %s is outside the file (%d lines)" % (os.path.relpath(f, ROOT), m.group(0), lengths[path]))
    assert n > 200, "citation pattern no longer matches anything"
    assert not bad, "\n".join(bad[:40])
