"""Turns a rocprofv3 --kernel-trace --stats result (rocpd sqlite .db) into a text table for profiles/.

    python tools/prof_summary.py gpurun_out/<dir> profiles/<name>.txt [note] [bygrid]

With "bygrid" every kernel is also listed per launch grid (the backward kernels serve layers of very different size).
"""
import glob
import sqlite3
import sys


def main():
    src, dst = sys.argv[1], sys.argv[2]
    note = sys.argv[3] if len(sys.argv) > 3 else ''
    by_grid = len(sys.argv) > 4 and sys.argv[4] == 'bygrid'
    dbs = sorted(glob.glob(src + '/**/*.db', recursive=True))
    lines = ['# rocprofv3 --kernel-trace --stats summary (%s)' % src, '# ' + note,
             '%-100s %8s %14s %12s %7s' % ('kernel', 'calls', 'total_us', 'avg_us', '%')]
    for db in dbs:
        con = sqlite3.connect(db)
        for name, calls, total, avg, pct in con.execute(
                'select name, total_calls, total_duration, average, percentage from top_kernels'):
            lines.append('%-100s %8d %14.1f %12.1f %7.2f' % (name[:100], calls, total, avg, pct))
    # the same kernel symbol serves layers of very different size: the launches of the dominant kernel per grid
    lines.append('#')
    lines.append('# launches of the 64 -> 64 convolution kernels by grid (threads x, y, z): the hot-path layers are the '
                 '48-plane ones; bench.py\'s roofline.launch_ms times one of them in isolation')
    lines.append('%-100s %22s %8s %12s' % ('kernel', 'grid', 'calls', 'avg_us'))
    for db in dbs:
        con = sqlite3.connect(db)
        for name, gx, gy, gz, calls, avg in con.execute(
                "select name, grid_x, grid_y, grid_z, count(*), avg(end - start) / 1000.0 from kernels "
                "where name like '%conv2d_wino%' or name like '%conv2d_x3%' or name like '%conv2d_mfma_kernel<4%' "
                "group by name, grid_x, grid_y, grid_z order by avg(end - start) desc"):
            lines.append('%-100s %22s %8d %12.1f' % (name[:100], '%dx%dx%d' % (gx, gy, gz), calls, avg))
    if by_grid:
        lines.append('#')
        lines.append('# every kernel above 20 us per launch, by grid (threads x, y, z)')
        lines.append('%-100s %22s %8s %12s %12s' % ('kernel', 'grid', 'calls', 'avg_us', 'total_us'))
        for db in dbs:
            con = sqlite3.connect(db)
            for name, gx, gy, gz, calls, avg, tot in con.execute(
                    "select name, grid_x, grid_y, grid_z, count(*), avg(end - start) / 1000.0, sum(end - start) / 1000.0 "
                    "from kernels group by name, grid_x, grid_y, grid_z having avg(end - start) > 20000 "
                    "order by sum(end - start) desc"):
                lines.append('%-100s %22s %8d %12.1f %12.1f' % (name[:100], '%dx%dx%d' % (gx, gy, gz), calls, avg, tot))
    open(dst, 'w').write('\n'.join(lines) + '\n')
    print('\n'.join(lines[:30]))


if __name__ == '__main__':
    main()
