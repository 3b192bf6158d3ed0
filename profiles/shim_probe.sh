cd /tmp && python - <<'PY'
import sys, time, subprocess, os
R=os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R)
from vsearch_amd import workload
flat, off, ln, fam = workload.make_family_db(20000, 1000, seed=17, device="cpu")
q, qo, ql, src = workload.make_queries(flat, off, ln, 2000, 250, seed=11, device="cpu")
b=flat.numpy().tobytes(); qb=q.numpy().tobytes()
open("db.fa","wb").write(b"".join(b">t%d\n%s\n"%(i,b[int(o):int(o)+int(l)]) for i,(o,l) in enumerate(zip(off,ln))))
open("q.fa","wb").write(b"".join(b">q%d\n%s\n"%(i,qb[int(o):int(o)+int(l)]) for i,(o,l) in enumerate(zip(qo,ql))))
env=dict(os.environ, VSX_DEBUG_TIMING="1", VSX_SHIM_STATS="1")
t0=time.time()
p=subprocess.run([R+"/oracle/_ref/vsearch_vsx","--usearch_global","q.fa","--db","db.fa","--id","0.9","--qmask","none","--dbmask","none","--threads","16","--userout","x.tsv","--userfields","query+target+id+caln","--quiet"],capture_output=True,text=True,env=env)
print("wall %.2f"%(time.time()-t0))
lines=[l for l in p.stderr.splitlines() if l.startswith("vsx_align_pairs:")]
import re
tot=[0,0,0,0]; n=0
for l in lines:
    m=re.search(r"plan ([\d.]+) s, run\+sync ([\d.]+) s, fetch ([\d.]+) s, destroy ([\d.]+) s", l)
    if m:
        n+=1
        for k in range(4): tot[k]+=float(m.group(k+1))
print(n, "calls; sums plan/run/fetch/destroy", [round(x,3) for x in tot])
print("\n".join(p.stderr.splitlines()[-6:]))
pl=[l for l in p.stderr.splitlines() if l.startswith("vsx_plan_create:")]
print(pl[5] if len(pl)>5 else pl)
PY
