#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2c16
mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $O/tr -o tr -- python $R/bench.py --no-extras --no-cpu-baseline --no-large-roofline --steps 1 --warmup 0 > $O/tr.log 2>&1
f=$(find $O/tr -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections, numpy as np
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
names=[r['Kernel_Name'] for r in rows]
st=np.array([int(r['Start_Timestamp']) for r in rows]); en=np.array([int(r['End_Timestamp']) for r in rows])
gz=np.array([int(r['Grid_Size_Z']) if 'Grid_Size_Z' in r else 0 for r in rows])
wz=np.array([int(r.get('Workgroup_Size_Z',1) or 1) for r in rows])
hess=[i for i,n in enumerate(names) if 'gmres_hess' in n]
seq=collections.defaultdict(list); cnt=0
for a,b in zip(hess[:-1],hess[1:]):
    if gz[a]//max(wz[a],1)==16 and b-a<=16:
        cnt+=1
        for pos,i in enumerate(range(a+1,b+1)):
            nm=names[i].split('(')[0].replace('void ','').replace('ricadi::','')[:46]
            seq[(pos,nm)].append((en[i]-st[i])/1e3)
print(cnt,"iterations at 16 groups")
tot=0
for k in sorted(seq):
    v=np.array(seq[k])
    if len(v)>cnt*0.5:
        print("%2d %-48s %7.1f us"%(k[0],k[1],np.median(v))); tot+=np.median(v)
print("sum",tot)
PY
rm -rf $O/tr
