#!/bin/bash
# round 6, call 19: k_wy_w / k_wy_update per launch shape (one stream, so that a duration is the kernel's own): where the
# averages of the statistics come from
mkdir -p gpurun_out/r06
R=${GRAFT_REPO_ROOT:-/root/repo}
out=/tmp/wyshape
( cd /tmp && export TMPDIR=/tmp && OGSQP_WIDE_AHEAD=0 timeout 900 rocprofv3 --kernel-trace --output-format csv -d $out -o b -- python $R/tools/sqp_solve.py launch4 3 1e-6 hip > $out.log 2>&1 )
f=$(ls $out/*kernel_trace.csv | head -1)
python - "$f" <<'PY' | tee $R/gpurun_out/r06/wy_shapes.txt
import csv,sys,re,collections
acc=collections.defaultdict(lambda:[0,0.0])
for r in csv.DictReader(open(sys.argv[1])):
    nm=r["Kernel_Name"]
    m=re.search(r"(k_wy_w|k_wy_update<\d)", nm)
    if not m: continue
    gx=int(r["Grid_Size_X"])//int(r["Workgroup_Size_X"]); gy=int(r["Grid_Size_Y"])//max(1,int(r["Workgroup_Size_Y"]))
    d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
    # bucket the row tiles
    key=(m.group(1), gx if gx<=4 else (gx//16)*16, gy)
    acc[key][0]+=1; acc[key][1]+=d
tot=collections.defaultdict(float)
for (k,gx,gy),(n,t) in sorted(acc.items()):
    tot[k]+=t
for (k,gx,gy),(n,t) in sorted(acc.items()):
    if t/tot[k]>0.02: print("%-14s row tiles ~%4d slices %3d: %5d launches, avg %8.1f us, %5.1f%% of this kernel's time"%(k,gx,gy,n,t/n,100*t/tot[k]))
PY
