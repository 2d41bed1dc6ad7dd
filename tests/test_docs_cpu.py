"""Guards the documents against rot: every repository path DESIGN.md, INTEGRATION.md, README.md and
profiles/README.md cite in backticks must exist (reference paths, globs and placeholders are skipped)."""
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
