import numpy as np, sys
sys.path.insert(0,'/root/repo')
from vision3d_amd import synth
pts=synth.make_cloud(0,16384)[:,:3].astype(np.float32)
N=len(pts); K=2048
def run(J):
    td=np.full(N,1e10,np.float32)
    sel=[0]; rounds=0
    cur=[0]
    while len(sel)<K:
        for c in cur:
            d=((pts-pts[c])**2)
            d=(d[:,0]+d[:,1])+d[:,2]
            td=np.minimum(td,d)
        rounds+=1
        # top-J by (d desc, idx asc)
        order=np.lexsort((np.arange(N),-td))[:J]
        acc=[order[0]]
        for c in order[1:]:
            if td[c]<=0: break
            ok=True
            for a in acc:
                dd=(pts[c]-pts[a])**2; dd=(dd[0]+dd[1])+dd[2]
                if dd<td[c]: ok=False;break
            if not ok: break
            acc.append(c)
        acc=acc[:K-len(sel)]
        sel+=list(acc); cur=acc
    return sel,rounds
ref,_=run(1)
for J in (2,3,4,6,8):
    s,r=run(J)
    print(J, 'rounds',r,'steps/round',(K-1)/r, 'exact', s==ref)
