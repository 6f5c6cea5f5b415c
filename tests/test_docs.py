"""DESIGN.md stays a document: lines wrapped at 150 columns (table rows excepted: markdown cannot wrap them), no table cell over 400
characters, every profiles/r06_* file the text names exists, the history it points to is there, and the sections a reader looks for are present."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TEXT = open(os.path.join(ROOT, "DESIGN.md")).read()


def test_line_lengths_and_table_cells():
    for n, line in enumerate(TEXT.split("\n"), 1):
        if line.startswith("|"):
            for cell in line.strip("|").split("|"):
                assert len(cell.strip()) <= 400, "DESIGN.md:%d: a table cell of %d characters" % (n, len(cell.strip()))
        else:
            assert len(line) <= 150, "DESIGN.md:%d: %d columns" % (n, len(line))
    assert len(TEXT) < 120_000                     # (203 kB at the end of round 5)


def test_every_named_profile_exists():
    names = set(re.findall(r"profiles/(r06_[A-Za-z0-9_.{},-]+)", TEXT))
    assert names, "DESIGN.md cites no round-6 profile"
    for nm in names:
        nm = nm.rstrip(".,)")
        if "{" in nm:                                # profiles/r06_pmc_{fetch,write}_size.csv
            head, rest = nm.split("{", 1); alts, tail = rest.split("}", 1)
            expand = [head + a + tail for a in alts.split(",")]
        else:
            expand = [nm]
        for f in expand:
            assert os.path.exists(os.path.join(ROOT, "profiles", f)), "DESIGN.md names profiles/%s, which does not exist" % f


def test_structure():
    for heading in ("## 1. The path and its boundary", "## 2. Oracle", "## 3. Data layout", "## 4. Launch structures", "### 4.1 The persistent solve", "### 4.3 Co-residency guards and the fallback ladder",
                    "## 5. Kernel roles", "## 6. Measurement", "## 7. Multi-GPU", "## 9. Out of scope", "## Appendix A: measured and not kept"):
        assert heading in TEXT, heading
    assert "PARITY UNPINNED" in TEXT
    assert os.path.exists(os.path.join(ROOT, "docs", "history", "DESIGN_rounds1-5.md"))
    for f in re.findall(r"`(tests/[a-z_0-9]+\.py)`", TEXT) + ["tests/" + t for t in re.findall(r"`(test_[a-z_0-9]+\.py)", TEXT)]:
        assert os.path.exists(os.path.join(ROOT, f)), f
