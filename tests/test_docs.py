"""INTEGRATION.md must not drift from the headers: every C-ABI function, shim method and vil:: name its code blocks use exists
in include/; every file:line style path it or DESIGN.md cites under this repository exists."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_integration_md_names_exist_in_headers():
    hdr = "".join(open(os.path.join(ROOT, "include", f)).read() for f in sorted(os.listdir(os.path.join(ROOT, "include"))))
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    names = set(re.findall(r"\b((?:vil|vgicp|vmap|vpre)_[a-z_0-9]+)\s*\(", doc))
    names |= set(re.findall(r"\b(?:pk|feats_|frames_|vil_prior_)\.([a-z_0-9]+)\s*\(", doc))
    names |= set(re.findall(r"vil::([A-Za-z_0-9]+)", doc))
    assert len(names) > 30
    missing = [n for n in sorted(names) if not re.search(r"\b" + re.escape(n) + r"\b", hdr)]
    assert not missing, missing


def test_repository_paths_cited_in_the_docs_exist():
    pat = re.compile(r"`((?:include|tests|tools|oracle|profiles|examples|mvil-fusion_amd)/[A-Za-z0-9_./-]+\.(?:h|hpp|hip|py|md|json|csv|txt|npz|cpp|sh))`")
    for doc in ("DESIGN.md", "INTEGRATION.md", "README.md", os.path.join("profiles", "README.md")):
        text = open(os.path.join(ROOT, doc)).read()
        for path in set(pat.findall(text)):
            assert os.path.exists(os.path.join(ROOT, path)), "%s cites %s" % (doc, path)
