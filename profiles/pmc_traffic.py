#!/usr/bin/env python
"""HBM-side traffic per launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) -> profiles/pmc_traffic.json.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out_f -o f -- python bench.py --no-graph --steps 3 --warmup 1 --no-cpu-baseline
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d out_w -o w -- python bench.py --no-graph --steps 3 --warmup 1 --no-cpu-baseline
    python profiles/pmc_traffic.py out_f/f_results.db out_w/w_results.db profiles/pmc_traffic.json

The counters are memory-side request tallies (TCC_EA*), reported in KiB.  MI355X_MICROARCH.md notes that on gfx950
FETCH_SIZE under-reports wide coalesced reads by 2x and that WRITE_SIZE is uncalibrated, so both are CALIBRATED here on
a kernel of the same run whose traffic is known exactly: sgd_kernel streams 12 B/parameter in and 8 B/parameter out
(float4 accesses over the 24.3M-parameter flat buffers).  The scale factors are stored next to the results."""
import hashlib
import json
import os
import sqlite3
import sys

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "d3feat.pytorch_amd", "csrc")
# kernel -> the source file it lives in (the file's SHA-256 is stored with the counters so that bench.py can tell a
# traffic figure measured on an older kernel from a current one)
KERNEL_FILES = {"kpconv_bwd_dx_kernel": "kpconv_fused.hip", "kpconv_fwd_fused_kernel": "kpconv_fused.hip",
                "kpconv_dx_gather_kernel": "kpconv_dx_gather.hip", "atb_partial_kernel": "linear.hip",
                "kpconv_agg_fwd_kernel": "kpconv_aggregate.hip", "kpconv_agg_rev_kernel": "kpconv_aggregate.hip",
                "rowgemm_kernel": "linear.hip", "atb_grouped_kernel": "linear.hip",
                "atb_grouped_reduce_kernel": "linear.hip",
                "bias_act_bwd_kernel": "elementwise.hip", "bias_act_fwd_kernel": "elementwise.hip",
                "pack_supports_kernel": "kpconv_fused.hip", "radius_query_kernel": "radius_neighbors.hip",
                "order_kernel": "grid_subsample.hip"}


def source_sha(kernel):
    path = os.path.join(CSRC, KERNEL_FILES[kernel])
    with open(path, "rb") as fh:
        return hashlib.sha256(fh.read()).hexdigest()[:16]


def per_kernel(db_path, counter):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    pmc = [t for t in tabs if t.startswith('rocpd_pmc_event_')][0]
    info = [t for t in tabs if t.startswith('rocpd_info_pmc_')][0]
    kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch_')][0]
    ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol_')][0]
    q = ("select s.kernel_name, avg(p.value), count(*) from %s p join %s i on p.pmc_id = i.id join %s d on "
         "p.event_id = d.event_id join %s s on d.kernel_id = s.id where i.name = '%s' group by 1" % (pmc, info, kd, ks, counter))
    return {name: (val, n) for name, val, n in cur.execute(q)}


def pick(table, needle):
    tot = cnt = 0.0
    for name, (val, n) in table.items():
        if needle in name:
            tot += val * n
            cnt += n
    return (tot / cnt, int(cnt)) if cnt else (None, 0)


def main(fetch_db, write_db, out_path, n_params=24304993):
    fetch, write = per_kernel(fetch_db, "FETCH_SIZE"), per_kernel(write_db, "WRITE_SIZE")
    sgd_f, _ = pick(fetch, "sgd_kernel")
    sgd_w, _ = pick(write, "sgd_kernel")
    kib = 1024.0
    f_scale = (12.0 * n_params) / (sgd_f * kib) if sgd_f else None
    w_scale = (8.0 * n_params) / (sgd_w * kib) if sgd_w else None
    out = {"_calibration": {"kernel": "sgd_kernel", "n_params": n_params, "FETCH_SIZE_raw_KiB": sgd_f,
                            "WRITE_SIZE_raw_KiB": sgd_w, "fetch_scale": f_scale, "write_scale": w_scale,
                            "note": "bytes = raw KiB * 1024 * scale; scales from sgd_kernel's exactly known streams"}}
    for key in KERNEL_FILES:
        f, nf = pick(fetch, key)
        w, nw = pick(write, key)
        if f is None or w is None:
            continue
        fb, wb = f * kib * (f_scale or 1.0), w * kib * (w_scale or 1.0)
        out[key] = {"launches": nf, "FETCH_SIZE_raw_KiB": f, "WRITE_SIZE_raw_KiB": w, "fetch_bytes_per_launch": fb,
                    "write_bytes_per_launch": wb, "hbm_bytes_per_launch": fb + wb,
                    "source": KERNEL_FILES[key], "source_sha16": source_sha(key)}
    # the grouped weight gradient is a launch PAIR (all problems' tasks, then all slabs' sums): bench.py's roofline
    # object prices the pair, so its traffic is the two kernels' together
    if "atb_grouped_kernel" in out and "atb_grouped_reduce_kernel" in out:
        out["atb_grouped_kernel"]["hbm_bytes_per_launch_pair"] = (
            out["atb_grouped_kernel"]["hbm_bytes_per_launch"] + out["atb_grouped_reduce_kernel"]["hbm_bytes_per_launch"])
    with open(out_path, "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:4])
