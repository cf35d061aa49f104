#!/usr/bin/env python
"""Turn a rocprofv3 rocpd SQLite result (kernel trace) into the per-kernel stats CSV we commit under profiles/.

    python tools/rocpd_stats.py gpurun_out/prof_x/bench_results.db profiles/rNN_name_kernel_stats.csv
"""
import csv
import sqlite3
import sys


def main(db_path, out_path):
    db = sqlite3.connect(db_path)
    rows = db.execute("select name, total_calls, total_duration, average, percentage from top_kernels "
                      "order by total_duration desc").fetchall()
    with open(out_path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDuration(us)", "AverageDuration(us)", "Percentage"])
        for name, calls, tot, avg, pct in rows:
            w.writerow([name, calls, "%.3f" % tot, "%.3f" % avg, "%.3f" % pct])
    print("wrote %s (%d kernels)" % (out_path, len(rows)))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
