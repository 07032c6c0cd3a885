#!/usr/bin/env python3
"""Wrap the prose of a markdown file at WIDTH columns (default 120): paragraphs and list items are re-filled with a hanging indent;
tables, headings, fenced code and blank lines are left alone.    python scripts/reflow_md.py DESIGN.md [width]"""
import re
import sys
import textwrap

path = sys.argv[1]
width = int(sys.argv[2]) if len(sys.argv) > 2 else 120
lines = open(path).read().split("\n")
out, para, in_code = [], [], False
item = re.compile(r"^(\s*)([*\-+]|\d+\.)\s+")


def flush():
    global para
    if not para:
        return
    first = para[0]
    m = item.match(first)
    if m:
        lead = m.group(0)
        hang = " " * len(lead)
        text = " ".join([first[len(lead):].strip()] + [p.strip() for p in para[1:]])
        out.extend(textwrap.wrap(text, width=width, initial_indent=lead, subsequent_indent=hang, break_long_words=False, break_on_hyphens=False))
    else:
        ind = re.match(r"^\s*", first).group(0)
        text = " ".join(p.strip() for p in para)
        out.extend(textwrap.wrap(text, width=width, initial_indent=ind, subsequent_indent=ind, break_long_words=False, break_on_hyphens=False))
    para = []


for ln in lines:
    if ln.strip().startswith("```"):
        flush(); in_code = not in_code; out.append(ln); continue
    if in_code or not ln.strip() or ln.lstrip().startswith("|") or ln.startswith("#"):
        flush(); out.append(ln); continue
    if item.match(ln) and para:
        flush()
    para.append(ln)
flush()
open(path, "w").write("\n".join(out))
