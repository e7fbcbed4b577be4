"""Summarise rocprofv3 output databases (ROCm 7.2 writes rocpd sqlite files): per-kernel dispatch statistics of a
--kernel-trace run and per-kernel counter sums of --pmc runs.

    python tools/pmc_summary.py <dir> [kernel-name-substring ...]

Walks <dir> for *_results.db; prints, per database, the top kernels by total time (calls, average, total) and, where
counters were collected, the sum of every counter over all dispatches of each kernel whose name contains one of the
substrings (default: loop_).  The summaries kept under profiles/ are this script's output.
"""
from __future__ import annotations

import glob
import os
import sqlite3
import sys


def main() -> int:
    root = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out'
    pats = sys.argv[2:] or ['loop_']
    for db in sorted(glob.glob(os.path.join(root, '**', '*_results.db'), recursive=True)):
        con = sqlite3.connect(db)
        print(f'== {os.path.relpath(db, root)}')
        try:
            rows = con.execute('select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels '
                               'group by name order by sum(duration) desc limit 12').fetchall()
        except sqlite3.Error as e:
            print('   (no kernel table:', e, ')')
            rows = []
        tot = sum(r[2] for r in rows) or 1
        for name, n, s, a, mn, mx in rows:
            short = name if len(name) < 90 else name[:87] + '...'
            print(f'   {short:<90} calls {n:>5}  avg {a / 1e3:>12.1f} us  min {mn / 1e3:>10.1f}  max {mx / 1e3:>10.1f}  total {s / 1e6:>10.2f} ms  {100.0 * s / tot:5.1f} %')
        try:
            crow = con.execute('select kernel_name, counter_name, sum(value), count(distinct dispatch_id) from counters_collection '
                               'group by kernel_name, counter_name').fetchall()
        except sqlite3.Error:
            crow = []
        by = {}
        for kname, cname, val, nd in crow:
            if any(p in kname for p in pats):
                by.setdefault((kname, nd), {})[cname] = val
        for (kname, nd), cs in by.items():
            short = kname if len(kname) < 100 else kname[:97] + '...'
            print(f'   counters of {short} ({nd} dispatches):')
            for cname in sorted(cs):
                print(f'      {cname:<34} {cs[cname]:>20.0f}')
            if 'SQ_WAVE_CYCLES' in cs and cs['SQ_WAVE_CYCLES'] > 0:
                w = cs['SQ_WAVE_CYCLES']
                print(f'      -> wave-cycle split: parked(s_waitcnt/barrier) {100 * cs.get("SQ_WAIT_ANY", 0) / w:.1f} %  '
                      f'issue-stall {100 * cs.get("SQ_WAIT_INST_ANY", 0) / w:.1f} %  issuing {100 * cs.get("SQ_ACTIVE_INST_ANY", 0) / w:.1f} %')
            if cs.get('SQ_LDS_IDX_ACTIVE', 0) > 0:
                print(f'      -> LDS bank-conflict cycles / LDS active cycles = {cs.get("SQ_LDS_BANK_CONFLICT", 0) / cs["SQ_LDS_IDX_ACTIVE"]:.4f}')
            if 'FETCH_SIZE' in cs:
                print(f'      -> HBM read  {cs["FETCH_SIZE"] * 2 / 1024:.1f} MB total ({cs["FETCH_SIZE"] * 2 * 1024 / nd:.0f} B per dispatch; FETCH_SIZE KB x 2: gfx950 correction)')
            if 'WRITE_SIZE' in cs:
                print(f'      -> HBM write {cs["WRITE_SIZE"] / 1024:.1f} MB total ({cs["WRITE_SIZE"] * 1024 / nd:.0f} B per dispatch)')
        con.close()
    return 0


if __name__ == '__main__':
    sys.exit(main())
