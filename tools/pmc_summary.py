"""Per-kernel means of rocprofv3 --pmc counters (CSV output).

    python tools/pmc_summary.py gpurun_out/<dir> [kernel-substring]
"""
import collections
import csv
import glob
import sys


def main():
    src = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else ''
    sums = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.defaultdict(set)
    for path in glob.glob(src + '/**/*counter_collection.csv', recursive=True):
        for row in csv.DictReader(open(path)):
            name = row['Kernel_Name']
            if want not in name:
                continue
            sums[name][row['Counter_Name']] += float(row['Counter_Value'])
            calls[name].add(row['Dispatch_Id'])
    for name, counters in sums.items():
        n = len(calls[name])
        print('%s  (%d dispatches; per-dispatch means)' % (name[:110], n))
        for c, v in sorted(counters.items()):
            print('    %-32s %.4g' % (c, v / n))


if __name__ == '__main__':
    main()
