"""Summarise rocprofv3 output databases (ROCm 7.2 writes rocpd sqlite files): per-kernel dispatch statistics of a
--kernel-trace run and per-kernel counter sums of --pmc runs.

    python tools/pmc_summary.py <dir> [kernel-name-substring ...]
    python tools/pmc_summary.py <dir> --json profiles/rNN_pmc.json     # + the per-launch counter numbers bench.py attaches to its roofline objects

Walks <dir> for *_results.db; prints, per database, the top kernels by total time (calls, average, total) and, where
counters were collected, the sum of every counter over all dispatches of each kernel whose name contains one of the
substrings (default: loop_).  The summaries kept under profiles/ are this script's output.
"""
from __future__ import annotations

import glob
import os
import sqlite3
import sys


def write_json(root: str, out: str) -> None:
    """profiles/rNN_pmc.json for bench.py: per BASELINE config (the directories <what>_c<N> of tools/profile_round.sh) the dominant loop kernel's HBM
    bytes and MFMA-busy cycles PER LAUNCH and its kernel-trace average, stamped with the commit and the hash of the kernel sources it ran on."""
    import json
    import subprocess
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
    import bench
    cfgs = {}
    for c in (1, 2, 4):
        rec = {}
        for what in ('kt', 'fetch', 'write', 'inst', 'sq'):
            dbs = glob.glob(os.path.join(root, f'{what}_c{c}', '**', '*_results.db'), recursive=True)
            if not dbs:
                continue
            con = sqlite3.connect(dbs[0])
            try:
                if what == 'kt':
                    r = con.execute("select name, count(*), avg(duration) from kernels where name like '%loop_%' group by name order by sum(duration) desc limit 1").fetchone()
                    if r:
                        rec.update(kernel=r[0], kt_calls=r[1], kt_avg_ms=r[2] / 1e6)
                else:
                    rows = con.execute("select kernel_name, counter_name, sum(value), count(distinct dispatch_id) from counters_collection "
                                       "where kernel_name like '%loop_%' group by kernel_name, counter_name").fetchall()
                    top = max({k for k, _, _, _ in rows}, key=lambda k: sum(v for kk, _, v, _ in rows if kk == k), default=None) if what != 'fetch' else None
                    for kname, cname, val, nd in rows:
                        if top is not None and kname != top:
                            continue
                        rec.setdefault('kernel', kname)
                        if cname == 'FETCH_SIZE':
                            rec['fetch_bytes_per_launch'] = int(val * 2 * 1024 / nd)        # KB x 2: the gfx950 correction of MI355X_MICROARCH.md
                        elif cname == 'WRITE_SIZE':
                            rec['write_bytes_per_launch'] = int(val * 1024 / nd)
                        elif cname == 'SQ_VALU_MFMA_BUSY_CYCLES':
                            rec['mfma_busy_cycles_per_launch'] = int(val / nd)
                        elif cname == 'SQ_INSTS_MFMA':
                            rec['insts_mfma_per_launch'] = int(val / nd)
                        elif cname in ('SQ_WAVE_CYCLES', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_LDS_BANK_CONFLICT', 'SQ_LDS_IDX_ACTIVE'):
                            rec[cname] = int(val / nd)
            except sqlite3.Error:
                pass
            con.close()
        rec.setdefault('fetch_bytes_per_launch', 0)
        rec.setdefault('write_bytes_per_launch', 0)
        rec.setdefault('mfma_busy_cycles_per_launch', 0)
        if 'kernel' in rec:
            cfgs[str(c)] = rec
    commit = os.environ.get('COMMIT', '')
    if not commit:
        try:
            commit = subprocess.check_output(['git', 'log', '-1', '--format=%h'], text=True, stderr=subprocess.DEVNULL).strip()
        except (OSError, subprocess.CalledProcessError):
            commit = 'unknown'
    with open(out, 'w') as f:
        json.dump(dict(commit=commit, csrc_sha=bench.csrc_sha(), tool='tools/profile_round.sh + tools/pmc_summary.py --json', configs=cfgs), f, indent=1)
    print(f'wrote {out}: commit {commit}, kernel sources {bench.csrc_sha()}, configs {sorted(cfgs)}')


def main() -> int:
    root = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out'
    if '--json' in sys.argv:
        write_json(root, sys.argv[sys.argv.index('--json') + 1])
        return 0
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
