"""Rewrap a markdown file at <= 140 columns: paragraphs and list items are re-flowed (continuation lines keep the item's
indent), tables with a row longer than the limit become one block per row (**first cell** -- header: cell; ...), fenced code is
left alone."""
import re
import sys
import textwrap

LIMIT = 132


def wrap_block(first_prefix, rest_prefix, text):
    return textwrap.fill(" ".join(text.split()), width=LIMIT, initial_indent=first_prefix, subsequent_indent=rest_prefix,
                         break_long_words=False, break_on_hyphens=False)


def split_row(line):
    cells = [c.strip() for c in line.strip().strip("|").split("|")]
    return cells


def main(path):
    lines = open(path).read().split("\n")
    out = []
    i = 0
    in_code = False
    while i < len(lines):
        ln = lines[i]
        if ln.strip().startswith("```"):
            in_code = not in_code
            out.append(ln)
            i += 1
            continue
        if in_code:
            out.append(ln)
            i += 1
            continue
        # table
        if ln.startswith("|") and i + 1 < len(lines) and re.match(r"^\|[\s:|-]+\|\s*$", lines[i + 1]):
            j = i
            rows = []
            while j < len(lines) and lines[j].startswith("|"):
                rows.append(lines[j])
                j += 1
            if max(len(r) for r in rows) <= LIMIT + 2:
                out.extend(rows)
            else:
                hdr = split_row(rows[0])
                for r in rows[2:]:
                    cells = split_row(r)
                    head = cells[0] if cells else ""
                    out.append(wrap_block("* ", "  ", f"**{head}**"))
                    for h, c in zip(hdr[1:], cells[1:]):
                        if c:
                            out.append(wrap_block("  - ", "    ", f"*{h}:* {c}"))
            i = j
            continue
        # heading / blank / html
        if ln.strip() == "" or ln.startswith("#") or ln.startswith("<"):
            out.append(ln)
            i += 1
            continue
        # list item or paragraph: gather continuation lines
        m = re.match(r"^(\s*)([-*+]|\d+\.)\s+", ln)
        if m:
            indent = m.group(1)
            marker = m.group(2)
            first_prefix = f"{indent}{marker} "
            rest_prefix = indent + " " * (len(marker) + 1)
            text = ln[m.end():]
            i += 1
            while i < len(lines) and lines[i].strip() != "" and not re.match(r"^\s*([-*+]|\d+\.)\s+", lines[i]) and not lines[i].startswith("#") and not lines[i].startswith("|") and not lines[i].strip().startswith("```"):
                text += " " + lines[i].strip()
                i += 1
            out.append(wrap_block(first_prefix, rest_prefix, text))
            continue
        indent = re.match(r"^(\s*)", ln).group(1)
        text = ln.strip()
        i += 1
        while i < len(lines) and lines[i].strip() != "" and not re.match(r"^\s*([-*+]|\d+\.)\s+", lines[i]) and not lines[i].startswith("#") and not lines[i].startswith("|") and not lines[i].strip().startswith("```"):
            text += " " + lines[i].strip()
            i += 1
        out.append(wrap_block(indent, indent, text))
    open(path, "w").write("\n".join(out))


if __name__ == "__main__":
    for p in sys.argv[1:]:
        main(p)
