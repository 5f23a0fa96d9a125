#!/usr/bin/env python3
"""Re-flow a markdown file to <= WIDTH columns: long paragraph / list lines are wrapped (continuation lines indented under the item), and
tables with a row beyond WIDTH become bullet lists ("**first cell** -- header: cell; header: cell ...").  Code fences are left alone.
    python scripts/wrap_md.py in.md out.md [width]"""
import re
import sys
import textwrap

src, dst = sys.argv[1], sys.argv[2]
W = int(sys.argv[3]) if len(sys.argv) > 3 else 150
lines = open(src).read().split("\n")
out, i, fence = [], 0, False


def wrap(text, first, rest):
    return textwrap.wrap(text, W, initial_indent=first, subsequent_indent=rest, break_long_words=False, break_on_hyphens=False) or [first.rstrip()]


def cells(row):
    row = row.strip()
    row = row[1:] if row.startswith("|") else row
    row = row[:-1] if row.endswith("|") else row
    return [c.strip() for c in re.split(r"(?<!\\)\|", row)]


while i < len(lines):
    ln = lines[i]
    if ln.lstrip().startswith("```"):
        fence = not fence
        out.append(ln)
        i += 1
        continue
    if fence:
        out.append(ln)
        i += 1
        continue
    if ln.lstrip().startswith("|") and i + 1 < len(lines) and re.match(r"^\s*\|?\s*:?-{3,}", lines[i + 1]):
        j = i
        while j < len(lines) and lines[j].lstrip().startswith("|"):
            j += 1
        block = lines[i:j]
        if max(len(b) for b in block) <= W:
            out += block
        else:
            hdr = cells(block[0])
            for row in block[2:]:
                cs = cells(row)
                head = cs[0] if cs else ""
                rest = []
                for h, c in zip(hdr[1:], cs[1:]):
                    if c:
                        rest.append(f"*{h}:* {c}" if h else c)
                text = (f"**{head}**" if head and not head.startswith("**") else head) + (" -- " + "; ".join(rest) if rest else "")
                out += wrap(text.replace("\\|", "|"), "* ", "  ")
        i = j
        continue
    if ln.strip() == "" or ln.startswith("#") or ln.startswith(">") or ln.lstrip().startswith("|"):
        out.append(ln)
        i += 1
        continue
    # a paragraph or list item: gather its continuation lines (same block: non-empty, not a new item / heading / table / fence), re-flow
    item = re.compile(r"^(\s*)([*+-]|\d+\.)\s+")
    m = item.match(ln)
    first = m.group(0) if m else re.match(r"^\s*", ln).group(0)
    rest_ind = " " * len(first) if m else first
    text = [ln[len(first):].strip()]
    j = i + 1
    while j < len(lines):
        nx = lines[j]
        if nx.strip() == "" or nx.startswith("#") or nx.lstrip().startswith("|") or nx.lstrip().startswith("```") or item.match(nx) or nx.startswith(">"):
            break
        text.append(nx.strip())
        j += 1
    out += wrap(" ".join(t for t in text if t), first, rest_ind)
    i = j
open(dst, "w").write("\n".join(out))
print(dst, "max line", max(len(l) for l in out), "lines", len(out))
