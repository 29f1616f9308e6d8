"""Turns a rocprofv3 --kernel-trace --stats result (rocpd sqlite .db) into a text table for profiles/.

    python tools/prof_summary.py gpurun_out/<dir> profiles/<name>.txt [note]
"""
import glob
import sqlite3
import sys


def main():
    src, dst = sys.argv[1], sys.argv[2]
    note = sys.argv[3] if len(sys.argv) > 3 else ''
    dbs = sorted(glob.glob(src + '/**/*.db', recursive=True))
    lines = ['# rocprofv3 --kernel-trace --stats summary (%s)' % src, '# ' + note,
             '%-100s %8s %14s %12s %7s' % ('kernel', 'calls', 'total_us', 'avg_us', '%')]
    for db in dbs:
        con = sqlite3.connect(db)
        for name, calls, total, avg, pct in con.execute(
                'select name, total_calls, total_duration, average, percentage from top_kernels'):
            lines.append('%-100s %8d %14.1f %12.1f %7.2f' % (name[:100], calls, total, avg, pct))
    open(dst, 'w').write('\n'.join(lines) + '\n')
    print('\n'.join(lines[:30]))


if __name__ == '__main__':
    main()
