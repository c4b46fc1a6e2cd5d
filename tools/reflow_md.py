#!/usr/bin/env python3
"""Reflows the prose of a markdown file to a column limit (tables, headings, fenced code and placeholders stay as they are).  usage: reflow_md.py <file.md> [width=158]"""
import re
import sys
import textwrap

path = sys.argv[1]; width = int(sys.argv[2]) if len(sys.argv) > 2 else 158
lines = open(path).read().split("\n")
out, block, kind, fenced = [], [], None, False
item = re.compile(r"^(\s*)([*-]|\d+\.)\s+")
def flush():
    global block, kind
    if not block: return
    text = " ".join(l.strip() for l in block)
    if kind == "item":
        m = item.match(block[0]); ind = m.group(1); mark = m.group(2)
        body = " ".join([block[0][m.end():].strip()] + [l.strip() for l in block[1:]])
        out.extend(textwrap.fill(body, width, initial_indent=ind + mark + " ", subsequent_indent=ind + " " * (len(mark) + 1), break_long_words=False, break_on_hyphens=False).split("\n"))
    else:
        out.extend(textwrap.fill(text, width, break_long_words=False, break_on_hyphens=False).split("\n"))
    block, kind = [], None
for l in lines:
    if l.startswith("```"):
        flush(); fenced = not fenced; out.append(l); continue
    if fenced or l.startswith("|") or l.startswith("#") or not l.strip() or l.strip().isupper() and "_" in l:
        flush(); out.append(l); continue
    if item.match(l):
        flush(); block, kind = [l], "item"; continue
    if kind is None:
        block, kind = [l], "para"
    elif kind == "para" and l.startswith(" "):      # (an indented line behind a paragraph: keep as continuation)
        block.append(l)
    else:
        block.append(l)
flush()
open(path, "w").write("\n".join(out))
