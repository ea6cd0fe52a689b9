"""Re-wrap a markdown file at 118 columns (paragraphs and list items are joined, then wrapped; tables whose rows are longer than 200
characters become nested lists; code fences and headings are left alone).  usage: python tools/md_reflow.py FILE..."""
import re, sys, textwrap
W = 118


def wrap(text, first, indent):
    return textwrap.fill(re.sub(r'\s+', ' ', text).strip(), width=W, initial_indent=first, subsequent_indent=indent, break_long_words=False,
                         break_on_hyphens=False)


def split_row(line):
    cells, cur, tick = [], '', False
    for ch in line.strip():
        if ch == '`':
            tick = not tick
        if ch == '|' and not tick:
            cells.append(cur); cur = ''
        else:
            cur += ch
    cells.append(cur)
    cells = [c.strip() for c in cells]
    if cells and cells[0] == '':
        cells = cells[1:]
    if cells and cells[-1] == '':
        cells = cells[:-1]
    return cells


def special(l):
    return l.strip() == '' or l.startswith('#') or l.startswith('|') or l.startswith('```') or l.startswith('>')


def reflow(src):
    out, i, n = [], 0, len(src)
    while i < n:
        line = src[i]
        if line.startswith('```'):
            out.append(line); i += 1
            while i < n and not src[i].startswith('```'):
                out.append(src[i]); i += 1
            if i < n:
                out.append(src[i]); i += 1
            continue
        if line.startswith('|') and i + 1 < n and re.match(r'^\|[-| :]+\|$', src[i + 1].strip()):
            header = split_row(line); i += 2; rows = []
            while i < n and src[i].startswith('|'):
                rows.append(split_row(src[i])); i += 1
            if max(len(' | '.join(r)) for r in rows) < 200:
                out.append('| ' + ' | '.join(header) + ' |'); out.append('|' + '|'.join(['---'] * len(header)) + '|')
                out += ['| ' + ' | '.join(r) + ' |' for r in rows]
            else:
                for r in rows:
                    r = r + [''] * (len(header) - len(r))
                    if len(header) == 2:
                        out.append(wrap('**' + r[0] + '** — ' + r[1], '* ', '  '))
                    else:
                        out.append(wrap('**' + r[0] + '**', '* ', '  '))
                        out += [wrap(f'*{h}*: {c}', '  - ', '    ') for h, c in zip(header[1:], r[1:]) if c]
                out.append('')
            continue
        if special(line):
            out.append(line); i += 1; continue
        m = re.match(r'^(\s*)([*-] |\d+\. )?(.*)$', line)
        ind, bul, txt = m.group(1), m.group(2) or '', m.group(3)
        i += 1
        while i < n and not special(src[i]) and not re.match(r'^\s*([*-] |\d+\. )', src[i]):
            txt += ' ' + src[i].strip(); i += 1
        out.append(wrap(txt, ind + bul, ind + ' ' * len(bul)))
    return out


for path in sys.argv[1:]:
    text = open(path).read()
    open(path, 'w').write('\n'.join(reflow(text.split('\n'))))
    print(path, 'longest line', max(len(l) for l in open(path).read().split('\n')))
