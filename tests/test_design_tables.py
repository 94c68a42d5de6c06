"""DESIGN.md's measured tables are GENERATED from the tracked files under profiles/ (tools/design_tables.py): what the
document quotes is what the files hold."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_design_md_holds_exactly_the_generated_tables():
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import design_tables as dt
    text = open(os.path.join(ROOT, 'DESIGN.md')).read()
    assert dt.BEGIN in text and dt.END in text
    block = text[text.index(dt.BEGIN) + len(dt.BEGIN):text.index(dt.END)]
    assert block.strip() == dt.build().strip()
    # ... and the instruction counts bench.py multiplies with its kernel time are the same file's
    import json
    v = json.load(open(os.path.join(ROOT, 'profiles', 'r4_valu.json')))
    for pt in v['points']:
        assert '{:,.0f}'.format(pt['valu_insts_per_eval']) in block
