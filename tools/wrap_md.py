#!/usr/bin/env python3
"""Re-flow a markdown file to <= WIDTH columns (default 120).  A paragraph or list item with a line over the limit is re-flowed
as a whole (its indentation and bullet kept); fenced code and headings are left alone; a table with a row over the limit
becomes a list -- one item per row, `**first cell** -- header: cell; ...` -- because a table row cannot be wrapped.
usage: wrap_md.py FILE [WIDTH]"""
import re
import sys
import textwrap

path = sys.argv[1]
W = int(sys.argv[2]) if len(sys.argv) > 2 else 120
lines = open(path).read().split("\n")
out = []
i = 0
BULLET = re.compile(r"^(\s*)(([*+-]|\d+\.)\s+)")


def cells(row):
    row = row.strip().strip("|")
    return [c.strip() for c in re.split(r"(?<!\\)\|", row)]


def wrap(text, first, rest):
    return textwrap.wrap(text, width=W, initial_indent=first, subsequent_indent=rest, break_long_words=False, break_on_hyphens=False)


def starts_block(l):
    return (not l.strip()) or l.startswith("#") or l.startswith("|") or l.lstrip().startswith("```") or BULLET.match(l) or l.startswith(">")


while i < len(lines):
    l = lines[i]
    if l.lstrip().startswith("```"):
        out.append(l)
        i += 1
        while i < len(lines) and not lines[i].lstrip().startswith("```"):
            out.append(lines[i])
            i += 1
        if i < len(lines):
            out.append(lines[i])
            i += 1
        continue
    if l.startswith("|") and i + 1 < len(lines) and re.match(r"^\|[\s:|-]+\|\s*$", lines[i + 1]):
        j = i
        while j < len(lines) and lines[j].startswith("|"):
            j += 1
        block = lines[i:j]
        if max(len(b) for b in block) <= W:
            out += block
        else:
            head = cells(block[0])
            for row in block[2:]:
                c = cells(row)
                parts = [f"{h}: {v}" if h else v for h, v in zip(head[1:], c[1:]) if v and v != "—"]
                first = c[0] if c[0].startswith("**") else f"**{c[0]}**"
                out += wrap(first + " — " + "; ".join(parts), "* ", "  ")
        i = j
        continue
    if not l.strip() or l.startswith("#") or l.startswith(">"):
        out.append(l)
        i += 1
        continue
    # a paragraph or one list item: this line + the lines that continue it
    j = i + 1
    while j < len(lines) and not starts_block(lines[j]):
        j += 1
    block = lines[i:j]
    if max(len(b) for b in block) <= W:
        out += block
    else:
        m = BULLET.match(l)
        if m:
            first, rest = m.group(1) + m.group(2), m.group(1) + " " * len(m.group(2))
            text = " ".join([l[len(first):].strip()] + [b.strip() for b in block[1:]])
        else:
            ind = re.match(r"^\s*", l).group(0)
            first = rest = ind
            text = " ".join(b.strip() for b in block)
        out += wrap(re.sub(r"(?<=[.:;!?])  +", "  ", text), first, rest)
    i = j
open(path, "w").write("\n".join(out))
