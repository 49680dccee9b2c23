"""Re-wrap a markdown file to a maximum line length: long paragraph / bullet lines are wrapped (continuation lines
indented under their bullet), tables with a row beyond the limit become lists (one item per row, one sub-item per
column), code fences and HTML comments are left alone.  python tools/wrap_markdown.py IN OUT [WIDTH]"""
import re
import sys
import textwrap


def wrap_line(line, width):
    if len(line) <= width:
        return [line]
    m = re.match(r"^(\s*)((?:[*+-]|\d+\.)\s+)?", line)
    indent, bullet = m.group(1), m.group(2) or ""
    body = line[len(indent) + len(bullet):]
    return textwrap.wrap(body, width=width, initial_indent=indent + bullet, subsequent_indent=indent + " " * len(bullet),
                         break_long_words=False, break_on_hyphens=False)


def cells(row):
    parts = re.split(r"(?<!\\)\|", row.strip())
    return [c.strip() for c in parts[1:-1]]


def table_to_list(rows, width):
    head = cells(rows[0])
    out = []
    for r in rows[2:]:
        c = cells(r)
        if not any(c):
            continue
        out += wrap_line(f"* {head[0]}: {c[0]}" if head and head[0] else f"* {c[0]}", width)
        for k in range(1, len(c)):
            if c[k]:
                name = head[k] if k < len(head) and head[k] else f"column {k + 1}"
                out += wrap_line(f"  - {name}: {c[k]}", width)
    return out


def main():
    src, dst = sys.argv[1], sys.argv[2]
    width = int(sys.argv[3]) if len(sys.argv) > 3 else 150
    lines = open(src, encoding="utf-8").read().split("\n")
    out, i, fence = [], 0, False
    while i < len(lines):
        ln = lines[i]
        if ln.lstrip().startswith("```"):
            fence = not fence
        if fence or ln.lstrip().startswith("<!--"):
            out.append(ln)
            i += 1
            continue
        if ln.startswith("|"):
            j = i
            while j < len(lines) and lines[j].startswith("|"):
                j += 1
            rows = lines[i:j]
            if any(len(r) > width for r in rows) and len(rows) >= 2 and re.match(r"^\|[\s:|-]+\|$", rows[1].strip()):
                out += table_to_list(rows, width)
            else:
                out += rows
            i = j
            continue
        out += wrap_line(ln, width)
        i += 1
    open(dst, "w", encoding="utf-8").write("\n".join(out))
    worst = max((len(x) for x in out), default=0)
    print(f"{dst}: {len(out)} lines, longest {worst}")


if __name__ == "__main__":
    main()
