#!/usr/bin/env python
"""Whole-step unit utilisation from rocprofv3 --pmc passes over profiles/net_step_only.py: every dispatch of the run,
grouped by kernel family, counters summed and set against the summed GRBM_GUI_ACTIVE of the same pass.  Answers "which
unit fills up first when several steps run side by side" (train.PairLanes): a unit that is busy a fraction u of ONE
step's time saturates at about 1/u steps in flight.
    pmc_step_totals.py db1 [db2 ...]  (one database per --pmc pass; GRBM_GUI_ACTIVE in every pass)
Normalisation as in pmc_digest.py: SQ_* per shader-engine instance (32 SIMDs, 8 CUs), quad-cycles except
SQ_VALU_MFMA_BUSY_CYCLES; TA_BUSY_avr in cycles; FETCH_SIZE / WRITE_SIZE in KiB with the gfx950 corrections measured on
sgd_kernel in this repository (x2.00 / x1.00, profiles/pmc_traffic.json)."""
import collections
import re
import sqlite3
import sys


def family(name):
    n = re.sub(r'^_ZN\w*?\d+', '', name)
    if name.startswith('Cijk'):
        return 'library GEMM (Cijk_*)'
    m = re.search(r'(\d+)([a-z_0-9]+_kernel)', name)
    if m:
        return m.group(2)
    if 'at6native' in name or 'at::native' in name:
        return 'ATen elementwise / copy'
    return n[:40]


def load(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    pmc = [t for t in tabs if t.startswith('rocpd_pmc_event_')][0]
    info = [t for t in tabs if t.startswith('rocpd_info_pmc_')][0]
    kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch_')][0]
    ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol_')][0]
    # per dispatch and counter: mean over the instances the counter is reported for
    q = ("select s.kernel_name, i.name, d.dispatch_id, avg(p.value), d.end - d.start from %s p join %s i on p.pmc_id = i.id "
         "join %s d on p.event_id = d.event_id join %s s on d.kernel_id = s.id group by 1, 2, 3" % (pmc, info, kd, ks))
    out = collections.defaultdict(lambda: collections.defaultdict(float))
    for name, cname, _, val, dur in cur.execute(q):
        fam = family(name)
        out[fam][cname] += val
        out['TOTAL'][cname] += val
        if cname == 'GRBM_GUI_ACTIVE':
            out[fam]['_n'] += 1
            out['TOTAL']['_n'] += 1
    return out


def main(paths):
    passes = [load(p) for p in paths]
    fams = sorted(passes[0], key=lambda f: -passes[0][f].get('GRBM_GUI_ACTIVE', 0))
    tot_gui = passes[0]['TOTAL']['GRBM_GUI_ACTIVE']

    def ratio(fam, counter, norm):
        for ps in passes:
            v = ps.get(fam, {})
            if counter in v and v.get('GRBM_GUI_ACTIVE'):
                return v[counter] / (norm * v['GRBM_GUI_ACTIVE'])
        return None

    cols = [("mfma", 'SQ_VALU_MFMA_BUSY_CYCLES', 32.0), ("valu", 'SQ_ACTIVE_INST_VALU', 8.0), ("lds", 'SQ_ACTIVE_INST_LDS', 2.0),
            ("vmem", 'SQ_ACTIVE_INST_VMEM', 8.0), ("salu", 'SQ_ACTIVE_INST_SCA', 8.0), ("ta", 'TA_BUSY_avr', 1.0),
            ("sq_busy", 'SQ_BUSY_CYCLES', 1.0)]
    print("%-34s %6s %7s | " % ("kernel family", "n", "time") + " ".join("%7s" % c[0] for c in cols) + " | %6s %9s" % ("l2_hit", "HBM GB/s"))
    for fam in fams[:18]:
        v = passes[0][fam]
        line = "%-34s %6d %6.1f%% | " % (fam, v['_n'], 100.0 * v['GRBM_GUI_ACTIVE'] / tot_gui)
        for _, c, norm in cols:
            r = ratio(fam, c, norm)
            line += " %7s" % ("%.3f" % r if r is not None else "-")
        hit = miss = fetch = write = gui_f = gui_w = None
        for ps in passes:
            w = ps.get(fam, {})
            if 'TCC_HIT_sum' in w:
                hit, miss = w['TCC_HIT_sum'], w['TCC_MISS_sum']
            if 'FETCH_SIZE' in w:
                fetch, gui_f = w['FETCH_SIZE'], w['GRBM_GUI_ACTIVE']
            if 'WRITE_SIZE' in w:
                write, gui_w = w['WRITE_SIZE'], w['GRBM_GUI_ACTIVE']
        l2 = "%.3f" % (hit / (hit + miss)) if hit is not None and hit + miss else "-"
        gbs = "-"
        if fetch is not None and write is not None:
            clk = 2.4e9     # GUI cycles -> seconds
            gbs = "%.0f" % ((2.0 * fetch * 1024 / (gui_f / clk) + 1.0 * write * 1024 / (gui_w / clk)) / 1e9)
        print(line + " | %6s %9s" % (l2, gbs))


if __name__ == "__main__":
    main(sys.argv[1:])
