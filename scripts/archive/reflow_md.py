#!/usr/bin/env python3
"""Reflow a markdown file to <= WIDTH columns: paragraphs and list items are re-wrapped; a table any of whose cells is longer than
CELL_MAX characters is rewritten row by row as a bold lead-in + one indented paragraph per further column (tables of short cells
stay tables); code blocks are left alone.   python scripts/reflow_md.py in.md out.md [cell_max]"""
import re
import sys
import textwrap

WIDTH, CELL_MAX = 118, 140


def wrap(text, indent="", first=None):
    first = indent if first is None else first
    return textwrap.fill(" ".join(text.split()), width=WIDTH, initial_indent=first, subsequent_indent=indent, break_long_words=False,
                         break_on_hyphens=False)


def split_row(line):
    cells, cur, code = [], "", False
    s = line.strip()
    s = s[1:] if s.startswith("|") else s
    s = s[:-1] if s.endswith("|") else s
    i = 0
    while i < len(s):
        ch = s[i]
        if ch == "`":
            code = not code
        if ch == "\\" and i + 1 < len(s) and s[i + 1] == "|":
            cur += "|"
            i += 2
            continue
        if ch == "|" and not code:
            cells.append(cur.strip())
            cur = ""
        else:
            cur += ch
        i += 1
    cells.append(cur.strip())
    return cells


def table(block, out):
    rows = [split_row(l) for l in block]
    header, body = rows[0], rows[2:]
    if max((len(c) for r in body for c in r), default=0) <= CELL_MAX and (len(sys.argv) > 3 or max(len(l) for l in block) <= 2 * WIDTH):
        out.extend(block)
        return
    for r in body:
        lead = r[0] if r else ""
        out.append(wrap(f"**{lead}**" if lead and not lead.startswith("**") else lead))
        for h, c in zip(header[1:], r[1:]):
            if c:
                out.append(wrap(f"*{h}:* {c}" if h else c, indent="    ", first="  - "))
        out.append("")


def main():
    global CELL_MAX
    if len(sys.argv) > 3:
        CELL_MAX = int(sys.argv[3])
    src = open(sys.argv[1]).read().split("\n")
    out, i, para = [], 0, []

    def flush():
        if para:
            text = " ".join(p.strip() for p in para)
            m = re.match(r"^(\s*)([-*+]|\d+\.)\s+", para[0])
            if m:
                ind = " " * len(m.group(0))
                out.append(wrap(text[len(m.group(0).lstrip()):] if False else re.sub(r"^\s*([-*+]|\d+\.)\s+", "", text), indent=ind, first=m.group(0)))
            elif para[0].startswith(">"):
                out.append(wrap(re.sub(r"^>\s?", "", text), indent="> ", first="> "))
            else:
                out.append(wrap(text))
            para.clear()
    while i < len(src):
        line = src[i]
        if line.strip().startswith("```"):
            flush()
            out.append(line)
            i += 1
            while i < len(src) and not src[i].strip().startswith("```"):
                out.append(src[i])
                i += 1
            if i < len(src):
                out.append(src[i])
            i += 1
            continue
        if line.lstrip().startswith("|") and i + 1 < len(src) and re.match(r"^\s*\|[\s:|-]+\|\s*$", src[i + 1]):
            flush()
            block = []
            while i < len(src) and src[i].lstrip().startswith("|"):
                block.append(src[i])
                i += 1
            table(block, out)
            continue
        if not line.strip():
            flush()
            out.append("")
        elif line.startswith("#"):
            flush()
            out.append(line)
        elif re.match(r"^\s*([-*+]|\d+\.)\s+", line) or line.startswith(">"):
            flush()
            para.append(line)
        else:
            para.append(line)
        i += 1
    flush()
    text = re.sub(r"\n{3,}", "\n\n", "\n".join(out))
    open(sys.argv[2], "w").write(text.rstrip("\n") + "\n")


if __name__ == "__main__":
    main()
