#!/usr/bin/env python3
"""rocprofv3 (ROCm 7.2 default output: a rocpd SQLite database) -> the per-kernel summary that `--stats` prints, as CSV:
Name, Calls, TotalDurationNs, AverageNs, Percentage, MinNs, MaxNs -- committed under profiles/.

    rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o p -- python3 bench.py ...
    python tools/rocpd_stats.py gpurun_out/prof/p_results.db > profiles/rNN_kernel_stats.csv
"""
import csv
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                      "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows)
    w = csv.writer(sys.stdout)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for name, calls, tot, avg, mn, mx in rows:
        w.writerow([name, calls, int(tot), round(avg, 1), round(100.0 * tot / total, 4), int(mn), int(mx)])
    span = db.execute("select min(start), max(end) from kernels").fetchone()
    print(f"# kernels: {sum(r[1] for r in rows)} dispatches, {total / 1e6:.2f} ms of kernel time in a "
          f"{(span[1] - span[0]) / 1e6:.2f} ms span", file=sys.stderr)


if __name__ == "__main__":
    main()
