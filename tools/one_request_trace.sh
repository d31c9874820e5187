cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=$PWD
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/req1 -o req1 -- python $R/tools/one_request_trace.py --natural ${TRACE_ARGS:-} > $R/gpurun_out/req1.log 2>&1 )
echo rc=$?; tail -7 gpurun_out/req1.log
f=$(find gpurun_out/req1 -name "*kernel_trace.csv" | head -1); echo $f
python - "$f" <<'PY'
import csv,re,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"],r["e"]=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    r["n"]=re.sub(r"\(.*","",r["Kernel_Name"]).replace("og::","").replace("void ","")
rows.sort(key=lambda r:r["s"])
# last call: kernels after the last gap > 30 ms
cut=0
for i in range(1,len(rows)):
    if rows[i]["s"]-rows[i-1]["e"]>30e6: cut=i
last=rows[cut:]
t0=last[0]["s"]
for r in last:
    print(f'{r["Stream_Id"]:>2} {(r["s"]-t0)/1e3:8.1f} {(r["e"]-t0)/1e3:8.1f} {(r["e"]-r["s"])/1e3:7.1f} us {r["n"][:60]} grid {r["Grid_Size_X"]} wg {r["Workgroup_Size_X"]}')
PY
