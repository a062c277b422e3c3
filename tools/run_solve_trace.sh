cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in "c2 cg" "c3 bicgstabl2" "c4 cg"; do
  set -- $c
  rocprofv3 --kernel-trace -d $R/gpurun_out/strace/$1_$2 -o out --output-format csv -- python3 $R/tools/solve_trace.py $1 $2 > $R/gpurun_out/strace_$1_$2.log 2>&1
  python3 - $R/gpurun_out/strace/$1_$2 $1 $2 <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+"/**/out_kernel_trace.csv",recursive=True)[0]
rows=sorted(csv.DictReader(open(f)),key=lambda r:int(r["Start_Timestamp"]))
marks=[i for i,r in enumerate(rows) if r["Kernel_Name"].startswith("k_axpby") and int(r["Grid_Size_X"])<=256*64]
a,b=marks[-2],marks[-1]
t0=int(rows[a]["End_Timestamp"])
print("==",sys.argv[2],sys.argv[3],"kernels of the last solve; total span %.3f ms"%((int(rows[b]["Start_Timestamp"])-t0)/1e6))
import collections
agg=collections.OrderedDict()
for r in rows[a+1:b]:
    k=r["Kernel_Name"].split("(")[0][:48]
    agg.setdefault(k,[0,0.0]); agg[k][0]+=1; agg[k][1]+=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6
for k,(n,t) in agg.items(): print(f"   {k:50s} x{n:3d} {t:8.3f} ms")
PY
done
