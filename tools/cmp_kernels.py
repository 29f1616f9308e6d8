"""Side-by-side average kernel durations (us) of several tools/prof_summary.py tables.

    python tools/cmp_kernels.py a.txt b.txt ...
"""
import re
import sys


def load(path):
    out = {}
    for line in open(path):
        if line.startswith('#') or line.startswith('kernel'):
            if line.startswith('# launches'):
                break
            continue
        m = re.match(r'(.*?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$', line)
        if m:
            name = m.group(1).replace('void pds::', '').replace('pds::', '')
            name = re.sub(r'\(.*', '', name)
            out[name.strip()] = (int(m.group(2)), float(m.group(4)))
    return out


def main():
    tables = [load(p) for p in sys.argv[1:]]
    names = []
    for t in tables:
        for n in t:
            if n not in names:
                names.append(n)
    print('%-50s' % 'kernel' + ''.join('%18s' % p.split('/')[-1][:17] for p in sys.argv[1:]))
    for n in names:
        row = '%-50s' % n[:50]
        for t in tables:
            row += '%10.1f x%-6d' % (t[n][1], t[n][0]) if n in t else '%18s' % '-'
        print(row)


if __name__ == '__main__':
    main()
