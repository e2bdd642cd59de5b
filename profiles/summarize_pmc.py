#!/usr/bin/env python3
"""Per-kernel averages of the counters in a rocprofv3 --pmc results .db (rocpd sqlite `counters_collection` view).

    python profiles/summarize_pmc.py <results.db> [kernel-name substring]
Prints, per kernel and counter: dispatches, mean of the per-dispatch value (summed over counter instances)."""
import sqlite3
import sys


def main(path, sub=None):
    cur = sqlite3.connect(path).cursor()
    q = ("select kernel_name, counter_name, count(*), avg(v) from (select kernel_name, counter_name, dispatch_id, sum(value) as v "
         "from counters_collection group by kernel_name, counter_name, dispatch_id) group by kernel_name, counter_name order by avg(v) desc")
    print(f"# rocprofv3 --pmc summary of {path}: kernel | counter | dispatches | mean per dispatch")
    for name, ctr, n, v in cur.execute(q):
        if sub and sub not in name:
            continue
        print(f"  {name[:90]:<90} {ctr:<24} {n:>5} {v:>16.1f}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
