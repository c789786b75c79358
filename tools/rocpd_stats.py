#!/usr/bin/env python
"""Per-kernel statistics of a rocprofv3 run from its rocpd SQLite database (the default output of this ROCm's rocprofv3):
   python tools/rocpd_stats.py <results.db> [--csv out.csv] [--gaps]
name, calls, total / average / min / max duration (us), share of the summed kernel time; --gaps adds the idle time
between consecutive dispatches (launch boundaries) to the report."""
import argparse
import sqlite3


def main():
    p = argparse.ArgumentParser()
    p.add_argument('db')
    p.add_argument('--csv')
    p.add_argument('--gaps', action='store_true')
    p.add_argument('--skip', type=int, default=0, help='ignore the first N dispatches (warm-up)')
    a = p.parse_args()
    c = sqlite3.connect(a.db)
    rows = c.execute('select k.kernel_name, d.start, d.end, d.grid_size_x, d.workgroup_size_x from rocpd_kernel_dispatch d '
                     'join rocpd_info_kernel_symbol k on d.kernel_id = k.id order by d.start').fetchall()
    rows = rows[a.skip:]
    stats = {}
    for name, s, e, g, w in rows:
        name = name.split('(')[0]
        st = stats.setdefault(name, [0, 0, 1 << 62, 0, g // max(w, 1)])
        d = e - s
        st[0] += 1; st[1] += d; st[2] = min(st[2], d); st[3] = max(st[3], d)
    total = sum(v[1] for v in stats.values())
    out = ['"name","calls","total_us","avg_us","min_us","max_us","pct","workgroups"']
    for name, (n, t, mn, mx, wg) in sorted(stats.items(), key=lambda kv: -kv[1][1]):
        out.append('"{}",{},{:.1f},{:.2f},{:.2f},{:.2f},{:.2f},{}'.format(name, n, t / 1e3, t / n / 1e3, mn / 1e3, mx / 1e3, 100.0 * t / total, wg))
    if a.gaps and len(rows) > 1:
        gaps = [rows[i + 1][1] - rows[i][2] for i in range(len(rows) - 1)]
        gaps = [g for g in gaps if 0 <= g < 50000]
        out.append('"(idle between consecutive dispatches < 50 us)",{},{:.1f},{:.2f},{:.2f},{:.2f},,'.format(
            len(gaps), sum(gaps) / 1e3, sum(gaps) / len(gaps) / 1e3, min(gaps) / 1e3, max(gaps) / 1e3))
    text = '\n'.join(out)
    print(text)
    if a.csv:
        open(a.csv, 'w').write(text + '\n')


if __name__ == '__main__':
    main()
