#!/usr/bin/env python3
"""Re-wrap the prose of a Markdown file to <= WIDTH bytes per line (tables, code blocks and headings are left alone).
usage: reflow_md.py FILE [WIDTH=100]"""
import re
import sys

path = sys.argv[1]
width = int(sys.argv[2]) if len(sys.argv) > 2 else 100
lines = open(path, encoding="utf-8").read().split("\n")


def blen(s):
    return len(s.encode("utf-8"))


def wrap(words, first, rest):
    out, cur = [], first
    started = False
    for w in words:
        cand = cur + ("" if not started else " ") + w
        if started and blen(cand) > width:
            out.append(cur)
            cur = rest + w
        else:
            cur = cand
        started = True
    out.append(cur)
    return out


out, para, fence = [], [], False


def flush():
    global para
    if not para:
        return
    m = re.match(r"^(\s*)((?:[*\-]|\d+\.)\s+)?", para[0])
    indent, bullet = m.group(1), m.group(2) or ""
    first = indent + bullet
    rest = indent + " " * len(bullet)
    text = " ".join(l.strip() for l in para)
    text = text[len(bullet):] if bullet and text.startswith(bullet.strip()) else text
    out.extend(wrap(text.split(), first, rest))
    para = []


for l in lines:
    if l.strip().startswith("```"):
        flush(); fence = not fence; out.append(l); continue
    if fence or l.startswith("|") or l.startswith("#") or l.startswith("    ") and not para or not l.strip() or l.strip() == "---":
        flush(); out.append(l); continue
    if re.match(r"^\s*([*\-]|\d+\.)\s+", l) and para:
        flush()
    para.append(l)
flush()
open(path, "w", encoding="utf-8").write("\n".join(out))
