cd /tmp && python - <<'PY'
import sys, time, subprocess, os, random
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from vsearch_amd import workload
R=os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
flat, off, ln, fam = workload.make_family_db(20000, 1000, seed=17, device="cpu")
q, qo, ql, src = workload.make_queries(flat, off, ln, 2000, 250, seed=11, device="cpu")
b=flat.numpy().tobytes(); qb=q.numpy().tobytes()
open("db.fa","wb").write(b"".join(b">t%d\n%s\n"%(i,b[int(o):int(o)+int(l)]) for i,(o,l) in enumerate(zip(off,ln))))
open("q.fa","wb").write(b"".join(b">q%d\n%s\n"%(i,qb[int(o):int(o)+int(l)]) for i,(o,l) in enumerate(zip(qo,ql))))
for name in ("vsearch_ref","vsearch_vsx"):
    for th in (1,16):
        t0=time.time()
        p=subprocess.run([R+"/oracle/_ref/"+name,"--usearch_global","q.fa","--db","db.fa","--id","0.9","--qmask","none","--dbmask","none","--threads",str(th),"--userout",name+".tsv","--userfields","query+target+id+caln","--quiet"],capture_output=True,text=True)
        print(name, "threads", th, "rc", p.returncode, "wall %.2f s"%(time.time()-t0), p.stderr[-200:])
print("identical:", sorted(open("vsearch_ref.tsv").read().splitlines())==sorted(open("vsearch_vsx.tsv").read().splitlines()))
PY
