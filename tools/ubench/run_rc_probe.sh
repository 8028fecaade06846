set -e
mkdir -p gpurun_out/r2h
for b in 1 0; do
  LAB_NOCHECK=1 LAB_KSWEEP=0 LAB_BKC=$b timeout -k 10 120 tools/ubench/gemm_lab_nomath 64064 2>&1 | grep ksweep | sed "s/^/nomath bkc=$b /"
done
cd /tmp && export TMPDIR=/tmp
for b in 1 0; do
  LAB_NOCHECK=1 LAB_KSWEEP=0 LAB_BKC=$b timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2h/pmc_b$b -- $GRAFT_REPO_ROOT/tools/ubench/gemm_lab 64064 > $GRAFT_REPO_ROOT/gpurun_out/r2h/pmc_b$b.log 2>&1
done
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv,glob,collections
for b in (1,0):
    f=glob.glob(f'gpurun_out/r2h/pmc_b{b}/**/*counter_collection.csv',recursive=True)[0]
    tot=collections.defaultdict(float); n=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=(r['Kernel_Name'][:70],r['Counter_Name'])
        tot[k]+=float(r['Counter_Value']); n[k]+=1
    for k in sorted(tot): print(b,k,n[k],tot[k])
PY
