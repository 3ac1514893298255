#!/usr/bin/env python
"""Per-kernel summary (count / total / avg / min / max ns, percentage) from a rocprofv3 rocpd .db file,
in the same columns as `rocprofv3 --stats` kernel_stats.csv.  Usage: rocpd_summary.py results.db [out.csv]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = list(db.execute('select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) '
                           'from kernels group by name order by sum(end-start) desc'))
    tot = sum(r[2] for r in rows) or 1
    out = open(sys.argv[2], 'w') if len(sys.argv) > 2 else sys.stdout
    out.write('"Name","Calls","TotalDurationNs","AverageNs","MinNs","MaxNs","Percentage"\n')
    for r in rows:
        out.write('"%s",%d,%d,%.1f,%d,%d,%.2f\n' % (r[0].replace('"', "'"), r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / tot))


if __name__ == '__main__':
    main()
