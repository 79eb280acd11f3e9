"""The documents point at tests and at committed evidence by name: every `test_...` they name exists under tests/, every
`profiles/...` path and every file name in profiles/README.md exists (names with wildcards / braces are patterns and are skipped)."""
import glob
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DOCS = ["DESIGN.md", "README.md", "INTEGRATION.md", "profiles/README.md", "tools/README.md"]


def _tests_source():
    return "\n".join(open(f).read() for f in glob.glob(os.path.join(ROOT, "tests", "*.py")))


@pytest.mark.parametrize("doc", DOCS)
def test_named_tests_exist(doc):
    src = _tests_source()
    text = open(os.path.join(ROOT, doc)).read()
    names = set(re.findall(r"`(test_[A-Za-z0-9_]+)", text))
    missing = [n for n in sorted(names) if not re.search(r"def " + re.escape(n) + r"[A-Za-z0-9_]*\(", src)]
    assert not missing, missing


@pytest.mark.parametrize("doc", DOCS)
def test_named_profile_paths_exist(doc):
    text = open(os.path.join(ROOT, doc)).read()
    paths = set(re.findall(r"`(profiles/[A-Za-z0-9_@./<>*{},+-]+)`", text))
    missing = [p for p in sorted(paths) if not any(ch in p for ch in "*{<") and not os.path.exists(os.path.join(ROOT, p))]
    assert not missing, missing


def test_files_named_in_the_profiles_index_exist():
    text = open(os.path.join(ROOT, "profiles", "README.md")).read()
    names = set(re.findall(r"`([A-Za-z0-9_@.+-]+\.(?:json|txt|csv))`", text))
    missing = [n for n in sorted(names) if not os.path.exists(os.path.join(ROOT, "profiles", n))]
    assert not missing, missing
