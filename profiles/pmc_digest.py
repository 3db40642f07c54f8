"""Digest of the rocprofv3 --pmc passes over the network step (profiles/calls/r03_pmc_kpconv.sh -> pmc_table.py output):
per kernel instantiation the unit utilisations the counters imply, as a table and as profiles/r03_pmc_kpconv.json (read
by bench.py's roofline object).
    python profiles/pmc_digest.py gpurun_out/c13_pmc.txt profiles/r03_pmc_kpconv

Normalisation (MI355X_MICROARCH.md, rocprofv3 PMC): SQ_* values are per shader-engine instance (8 CUs = 32 SIMDs) and
count quad-cycles except SQ_VALU_MFMA_BUSY_CYCLES (cycles, summed over the instance's SIMDs); GRBM_GUI_ACTIVE = kernel
duration in cycles.  mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (32 x GUI); valu_busy = 4 x SQ_ACTIVE_INST_VALU / (32 x GUI);
lds_busy = 4 x SQ_ACTIVE_INST_LDS / (8 x GUI) (one LDS per CU); ta_busy = TA_BUSY_avr / GUI; l2_hit = TCC_HIT / (HIT + MISS);
wait_mem / wait_issue / issuing = SQ_WAIT_ANY / SQ_WAIT_INST_ANY / SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES."""
import collections
import json
import re
import sys

src, out = sys.argv[1], sys.argv[2]
cur, data = None, collections.OrderedDict()
for line in open(src):
    line = line.rstrip()
    if line.startswith(('+', '##')) or not line:
        continue
    if not line.startswith(' '):
        cur = line.strip()
        data.setdefault(cur, {})
        continue
    m = re.match(r'\s+(\S+)\s+([\d.]+)\s+\(n=(\d+)\)', line)
    if m and cur:
        data[cur][m.group(1)] = float(m.group(2))

KERNELS = {'kpconv_fwd_fused_kernel': 'kpconv_fwd_fused', 'kpconv_dx_gather_kernel': 'kpconv_dx_gather',
           'kpconv_bwd_dx_kernel': 'kpconv_bwd_dx', 'atb_partial_kernel': 'atb_partial',
           'kpconv_agg_fwd_kernel': 'kpconv_agg_fwd', 'kpconv_agg_rev_kernel': 'kpconv_agg_rev',
           'rowgemm_kernel': 'rowgemm', 'atb_grouped_kernel': 'atb_grouped_kernel',
           'atb_grouped_reduce_kernel': 'atb_grouped_reduce'}
rows, by_kernel = [], {}
for name, v in data.items():
    if not v or 'GRBM_GUI_ACTIVE' not in v:
        continue
    g = v['GRBM_GUI_ACTIVE']
    wc = v.get('SQ_WAVE_CYCLES', 0.0) or 1.0
    hit, miss = v.get('TCC_HIT_sum', 0.0), v.get('TCC_MISS_sum', 0.0)
    r = {"instantiation": name, "gui_active_cycles": round(g),
         "mfma_busy": round(v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (32 * g), 3),
         "valu_busy": round(4 * v.get('SQ_ACTIVE_INST_VALU', 0) / (32 * g), 3),
         "lds_busy": round(4 * v.get('SQ_ACTIVE_INST_LDS', 0) / (8 * g), 3),
         "ta_busy": round(v.get('TA_BUSY_avr', 0) / g, 3),
         "l2_hit": round(hit / (hit + miss), 3) if hit + miss else None,
         "wait_mem": round(v.get('SQ_WAIT_ANY', 0) / wc, 2), "wait_issue": round(v.get('SQ_WAIT_INST_ANY', 0) / wc, 2),
         "issuing": round(v.get('SQ_ACTIVE_INST_ANY', 0) / wc, 2),
         "lds_bank_conflict_share": round(v.get('SQ_LDS_BANK_CONFLICT', 0) / v['SQ_LDS_IDX_ACTIVE'], 3)
         if v.get('SQ_LDS_IDX_ACTIVE') else None}
    rows.append(r)
    for key, stem in KERNELS.items():
        if stem in name:
            by_kernel.setdefault(key, []).append(r)
with open(out + '.txt', 'w') as f:
    f.write(__doc__.split('Normalisation')[1].join(['# Normalisation', '']) if False else '')
    f.write("# unit utilisation of the hand-written kernels inside the network step (rocprofv3 --pmc, 5 passes; see pmc_digest.py for the normalisation)\n")
    f.write("%-44s %9s %9s %9s %8s %7s %8s %9s %10s %8s\n" % ("instantiation", "mfma_busy", "valu_busy", "lds_busy",
                                                             "ta_busy", "l2_hit", "wait_mem", "wait_issue", "issuing",
                                                             "lds_conf"))
    for r in rows:
        f.write("%-44s %9.3f %9.3f %9.3f %8.3f %7s %8.2f %9.2f %10.2f %8s\n" % (
            r["instantiation"][:44], r["mfma_busy"], r["valu_busy"], r["lds_busy"], r["ta_busy"], r["l2_hit"], r["wait_mem"],
            r["wait_issue"], r["issuing"], r["lds_bank_conflict_share"]))
summary = {}
for key, rs in by_kernel.items():
    # time-weighted over the instantiations
    tot = sum(r["gui_active_cycles"] for r in rs)
    summary[key] = {k: round(sum(r[k] * r["gui_active_cycles"] for r in rs if r[k] is not None) / tot, 3)
                    for k in ("mfma_busy", "valu_busy", "lds_busy", "ta_busy", "l2_hit", "wait_mem", "wait_issue")}
    summary[key]["source"] = ("rocprofv3 --pmc over profiles/net_step_only.py, time-weighted over the kernel's "
                              "instantiations (%s.txt)" % out)
with open(out + '.json', 'w') as f:
    json.dump(summary, f, indent=1)
print(open(out + '.txt').read())
print(json.dumps(summary, indent=1))
