set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/gt -o gt -f csv -- python profiles/gemm_trace_target.py 30 2>&1 | tail -5) > gpurun_out/c4_trace.log
ls -R gpurun_out/gt | head -20 >> gpurun_out/c4_trace.log
python - <<'PY' > gpurun_out/c4_kernels.txt 2>&1
import csv, glob, collections
f = glob.glob('gpurun_out/gt/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
seq = []
for r in rows:
    name = r['Kernel_Name']; d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1000.0
    g = (r.get('Grid_Size_X') or r.get('Grid_Size'), r.get('Workgroup_Size_X') or r.get('Workgroup_Size'))
    if seq and seq[-1][0] == name and seq[-1][3] == g: seq[-1][1].append(d)
    else: seq.append([name, [d], r.get('VGPR_Count', ''), g, r.get('LDS_Block_Size', '')])
for name, ds, vg, g, lds in seq:
    ds2 = sorted(ds)
    print("%-90.90s n=%3d med=%7.2f min=%7.2f grid=%s vgpr=%s lds=%s" % (name, len(ds), ds2[len(ds2)//2], ds2[0], g, vg, lds))
PY
rm -rf gpurun_out/gt
echo done
