"""Per-kernel means of rocprofv3 --pmc counters (rocpd sqlite output, or the older CSV output).

    python tools/pmc_summary.py gpurun_out/<dir> [kernel-substring]

Values are summed over the counter's dimensions (XCDs / shader engines) per dispatch, then averaged over the
dispatches of a kernel (grouped by kernel name AND grid, since one symbol serves layers of different size).
"""
import collections
import csv
import glob
import sqlite3
import sys


def main():
    src = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else ''
    sums = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.defaultdict(lambda: collections.defaultdict(set))
    durs = collections.defaultdict(dict)
    regs = {}
    for path in sorted(glob.glob(src + '/**/*.db', recursive=True)):
        con = sqlite3.connect(path)
        try:
            rows = con.execute('select kernel_name, grid_size_x, grid_size_y, grid_size_z, dispatch_id, counter_name, '
                               'value, duration, vgpr_count, accum_vgpr_count, sgpr_count, lds_block_size '
                               'from counters_collection')
        except sqlite3.Error:
            continue
        for name, gx, gy, gz, disp, counter, value, dur, vg, ag, sg, lds in rows:
            if want not in name:
                continue
            key = (name, gx, gy, gz)
            sums[key][counter] += float(value)
            calls[key][counter].add((path, disp))
            durs[key][(path, disp)] = dur
            regs[key] = (vg, ag, sg, lds)
    for path in glob.glob(src + '/**/*counter_collection.csv', recursive=True):
        for row in csv.DictReader(open(path)):
            name = row['Kernel_Name']
            if want not in name:
                continue
            key = (name, 0, 0, 0)
            sums[key][row['Counter_Name']] += float(row['Counter_Value'])
            calls[key][row['Counter_Name']].add((path, row['Dispatch_Id']))
    for key in sorted(sums, key=lambda k: -sum(durs[k].values()) if durs[k] else 0):
        name, gx, gy, gz = key
        d = durs[key]
        mean_us = sum(d.values()) / len(d) / 1e3 if d else 0.0
        print('%s  grid %dx%dx%d  (%.1f us mean under the profiler; vgpr %s agpr %s sgpr %s lds %s)' %
              ((name[:90], gx, gy, gz, mean_us) + tuple(regs.get(key, ('?',) * 4))))
        for c, v in sorted(sums[key].items()):
            print('    %-28s %.5g' % (c, v / len(calls[key][c])))


if __name__ == '__main__':
    main()
