#!/usr/bin/env python3
"""Re-wrap the prose of a markdown file at a column limit (default 118): paragraphs and list items are re-flowed (continuation lines of a list
item are indented under its text); headings, tables, fenced code and blank lines are left as they are.  tools/wrap_md.py FILE [WIDTH]"""
import re, sys, textwrap

def wrap(text, width):
    out, para, fence = [], [], False
    def flush():
        if not para:
            return
        first = para[0]
        m = re.match(r'^(\s*)((?:[-*+]|\d+\.)\s+)?', first)
        indent, bullet = m.group(1), m.group(2) or ''
        body = ' '.join([first[len(indent) + len(bullet):].strip()] + [l.strip() for l in para[1:]])
        sub = indent + ' ' * len(bullet)
        out.extend(textwrap.wrap(body, width=width, initial_indent=indent + bullet, subsequent_indent=sub, break_long_words=False, break_on_hyphens=False) or [indent + bullet])
        para.clear()
    for line in text.split('\n'):
        s = line.strip()
        if s.startswith('```'):
            flush(); fence = not fence; out.append(line); continue
        if fence or not s or s.startswith('#') or s.startswith('|') or s.startswith('{') or re.match(r'^[-=]{3,}$', s):
            flush(); out.append(line); continue
        m = re.match(r'^(\s*)((?:[-*+]|\d+\.)\s+)', line)
        if m and para:
            # a wrapped continuation line may begin with "35." or "- ": it is a new item only if it is not indented under the current one
            m0 = re.match(r'^(\s*)((?:[-*+]|\d+\.)\s+)?', para[0])
            under = len(m0.group(1)) + len(m0.group(2) or '')
            if m0.group(2) and len(m.group(1)) >= under:
                m = None
        if m:
            flush()
        para.append(line)
    flush()
    return '\n'.join(out)

if __name__ == '__main__':
    path = sys.argv[1]
    width = int(sys.argv[2]) if len(sys.argv) > 2 else 118
    src = open(path).read()
    open(path, 'w').write(wrap(src, width))
