"""Guards the documents against rot: every repository path DESIGN.md, INTEGRATION.md, README.md and
profiles/README.md cite in backticks must exist (reference paths, globs and placeholders are skipped), and so
must every repository file the public headers under include/ name in their comments."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PREFIXES = ("cudaraytracing_amd/", "oracle/", "tests/", "tools/", "profiles/", "include/", "examples/", "csrc/")


def cited_paths(doc):
    text = open(os.path.join(ROOT, doc)).read()
    out = set()
    for tok in re.findall(r"`([^`\n]+)`", text):
        tok = tok.split("::")[0].split(" ")[0].rstrip(".,;:")
        if not tok.startswith(PREFIXES) or any(c in tok for c in "*{}<>…$%|"):
            continue
        if not re.search(r"\.[A-Za-z0-9]+$", tok):
            continue
        out.add(tok)
    return sorted(out)


@pytest.mark.parametrize("doc", ["DESIGN.md", "INTEGRATION.md", "README.md", "profiles/README.md"])
def test_cited_files_exist(doc):
    missing = []
    for p in cited_paths(doc):
        rel = ("cudaraytracing_amd/" + p) if p.startswith("csrc/") else p
        if doc.startswith("profiles/") and not p.startswith(PREFIXES):
            rel = "profiles/" + p
        if p.startswith("oracle/_ref/") or p.endswith(".so"):
            continue                                  # built artefacts, not in the source tree
        if not os.path.exists(os.path.join(ROOT, rel)):
            missing.append(p)
    assert not missing, f"{doc} cites files that do not exist: {missing}"


@pytest.mark.parametrize("header", sorted(f for f in os.listdir(os.path.join(ROOT, "include")) if f.endswith((".h", ".hpp"))))
def test_headers_cite_only_files_that_exist(header):
    """A header's comments may name this repository's files (rc_*.h / rc_*.hpp headers, csrc/..., tests/...,
    tools/...); each must exist.  Names of the reference's files (Screen.h, kernel.cu ...) are not ours to check."""
    text = open(os.path.join(ROOT, "include", header)).read()
    names = set(re.findall(r"\b(?:include/)?(rc_[a-z0-9_]+\.(?:hpp|h))\b", text))
    names |= set(re.findall(r"\b((?:csrc|tests|tools|cudaraytracing_amd|oracle|examples)/[A-Za-z0-9_./]+\.[a-z]+)\b", text))
    missing = []
    for n in sorted(names):
        cands = [n, "include/" + n, "cudaraytracing_amd/" + n, "cudaraytracing_amd/csrc/" + n, "cudaraytracing_amd/csrc/host/" + n]
        if not any(os.path.exists(os.path.join(ROOT, c)) for c in cands):
            missing.append(n)
    assert not missing, f"include/{header} names files that do not exist: {missing}"
