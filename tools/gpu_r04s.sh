# which vendor kernel (macro tile, stream-K flags) runs which hot-path shape: kernel trace of gemm_bench --vendor, consecutive launches grouped
mkdir -p gpurun_out/r4s
export TMPDIR=/tmp; cd /tmp; (timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4s/prof -- python $GRAFT_REPO_ROOT/tools/gemm_bench.py --vendor --iters 10 > $GRAFT_REPO_ROOT/gpurun_out/r4s/bench.log 2> $GRAFT_REPO_ROOT/gpurun_out/r4s/prof.err); cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, re
f = glob.glob('gpurun_out/r4s/prof/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
print(list(rows[0].keys()))
out = []
prev = None
for r in rows:
    n = r['Kernel_Name']
    if not ('Cijk' in n or 'gemm_' in n): continue
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    g = lambda *ks: next((r[k] for k in ks if k in r), '?')
    key = (n, g('Grid_Size', 'Grid_Size_X'), g('Workgroup_Size', 'Workgroup_Size_X'), g('LDS_Block_Size', 'LDS_Block_Size_v'), g('VGPR_Count', 'Arch_VGPR_Count'), g('Accum_VGPR_Count'), g('SGPR_Count'))
    if prev and prev[0] == key: prev[1].append(d)
    else:
        prev = [key, [d]]; out.append(prev)
with open('gpurun_out/r4s/vendor_kernels.txt', 'w') as fo:
    for key, ds in out:
        n = key[0]
        m = re.search(r'MT\d+x\d+x\d+', n)
        short = (m.group(0) + (' SK3' if '_SK3' in n else '') + ' ' + ' '.join(re.findall(r'_(WG\d+_\d+_\d+|MIWT\d+_\d+|GSU\d+|PGR\d|PLR\d|1LDSB\d|LDSB\d|DTVA\d|DTVB\d|WGM\d+|SIA\d|NTA\d|NTB\d|TLDS\d|SU\d+|SUS\d+|LRVW\d+|LWPMn?\d+|WSGRA\d|WSGRB\d|CLR\d|NLCA\d|NLCB\d|GRVWA\d|GRVWB\d|WS\d+)(?=_)', n))) if m else n[:60]
        ds2 = sorted(ds)
        fo.write(f"{len(ds):3d}x  med {ds2[len(ds2)//2]:7.1f} us  grid {key[1]:>7s} wg {key[2]:>4s} lds {key[3]:>6s} vgpr {key[4]:>3s} agpr {key[5]:>3s} sgpr {key[6]:>3s}  {short}\n")
print(open('gpurun_out/r4s/vendor_kernels.txt').read())
PY
rm -rf gpurun_out/r4s/prof
