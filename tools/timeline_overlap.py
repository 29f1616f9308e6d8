"""How the three-stream schedule overlaps kernels: rocprofv3 --kernel-trace databases of a sequential and a pipelined run
of bench.py -> a text report (concurrency histogram, per-kernel duration sequential vs pipelined, time a kernel class runs
alone / beside others).

    python tools/timeline_overlap.py SEQ_TRACE_DIR PIPE_TRACE_DIR OUT.txt

Caveat (measured, round 5): under --kernel-trace the dispatches are close to serialised -- two kernels were in flight for
4 % of the three-stream run, none for 37 % -- so the report shows per-kernel slowdowns and the sum of kernel durations
(2.45 ms per pair against 2.74 ms of sequential wall time: 0.29 ms of launch-to-launch gaps), NOT the overlap of the
un-profiled schedule.
"""
import glob
import re
import sqlite3
import sys


def load(src):
    rows = []
    for db in sorted(glob.glob(src + '/**/*.db', recursive=True)):
        con = sqlite3.connect(db)
        rows += list(con.execute('select name, start, end from kernels'))
    rows.sort(key=lambda r: r[1])
    return rows


def short(name):
    name = re.sub(r'^void ', '', name)
    name = re.sub(r'\(.*$', '', name)
    name = name.replace('pds::', '').replace('(anonymous namespace)::', '')
    return name[:60]


def steady(rows, lo=0.35, hi=0.9):
    """the timed part: kernels whose start lies in [lo, hi] of the run (set-up and the parity legs are outside)"""
    t0, t1 = rows[0][1], max(r[2] for r in rows)
    a, b = t0 + lo * (t1 - t0), t0 + hi * (t1 - t0)
    return [r for r in rows if a <= r[1] <= b]


def concurrency(rows):
    ev = []
    for _, s, e in rows:
        ev.append((s, 1))
        ev.append((e, -1))
    ev.sort()
    hist, cur, last = {}, 0, ev[0][0]
    for t, d in ev:
        hist[cur] = hist.get(cur, 0) + (t - last)
        cur += d
        last = t
    return hist


def alone_time(rows):
    """per kernel name: time it runs with no other kernel in flight / with others"""
    ev = []
    for i, (_, s, e) in enumerate(rows):
        ev.append((s, 1, i))
        ev.append((e, -1, i))
    ev.sort()
    active, last = set(), ev[0][0]
    alone, shared = {}, {}
    for t, d, i in ev:
        dt = t - last
        if dt > 0 and active:
            for j in active:
                nm = short(rows[j][0])
                (alone if len(active) == 1 else shared)[nm] = (alone if len(active) == 1 else shared).get(nm, 0) + dt
        if d > 0:
            active.add(i)
        else:
            active.discard(i)
        last = t
    return alone, shared


def stats(rows):
    by = {}
    for n, s, e in rows:
        by.setdefault(short(n), []).append(e - s)
    return {k: (len(v), sum(v) / len(v)) for k, v in by.items()}


def main():
    seq, pipe, dst = load(sys.argv[1]), load(sys.argv[2]), sys.argv[3]
    seq_s, pipe_s = steady(seq), steady(pipe)
    out = ['# timeline of the hot path under rocprofv3 --kernel-trace: sequential pairs vs the default three-stream schedule',
           '# (kernels whose start lies in the middle of each run; durations in us)']
    for label, rows in (('sequential', seq_s), ('three streams', pipe_s)):
        wall = max(r[2] for r in rows) - rows[0][1]
        h = concurrency(rows)
        tot = sum(h.values())
        out.append('%s: %d kernels over %.1f ms; sum of durations %.1f ms; kernels in flight: %s' % (
            label, len(rows), wall / 1e6, sum(r[2] - r[1] for r in rows) / 1e6,
            ', '.join('%d: %.1f %%' % (k, 100.0 * v / tot) for k, v in sorted(h.items()))))
    a, b = stats(seq_s), stats(pipe_s)
    alone, shared = alone_time(pipe_s)
    out.append('')
    out.append('%-62s %6s %10s %10s %7s %12s %12s' % ('kernel', 'calls', 'seq us', 'piped us', 'ratio', 'alone ms', 'beside ms'))
    for k, (n, avg) in sorted(b.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
        if n * avg < 0.5e6:   # below 0.5 ms in total
            continue
        sa = a.get(k, (0, 0.0))[1]
        out.append('%-62s %6d %10.1f %10.1f %7.2f %12.2f %12.2f' % (k, n, sa / 1e3, avg / 1e3, avg / sa if sa else 0.0,
                                                                   alone.get(k, 0) / 1e6, shared.get(k, 0) / 1e6))
    open(dst, 'w').write('\n'.join(out) + '\n')
    print('\n'.join(out))


if __name__ == '__main__':
    main()
