import sys; sys.path.insert(0,'.')
import numpy as np
from oracle import oracle as O
from rfsurfhmc_amd.model.lib import libsurf
g=np.load('tests/golden/swd_reference.npz')
rel=lambda a,b: np.abs(a-b).max()/max(np.abs(b).max(),1e-300)
names=sorted({k.split('/')[0] for k in g.files})
for name in names:
    thk,vs,t=g[name+'/thk'],g[name+'/vs'],g[name+'/t']
    vp,rho,_,_=O.empirical_relation(vs)
    for wt in ('Rc','Rg'):
        if f'{name}/{wt}/c' not in g.files: continue
        r=libsurf.adjoint_kernel(thk,vp,vs,rho,t,wt)
        if not r[5]: print(name,wt,'fail',bool(g[f'{name}/{wt}/flag'])); continue
        ref=[g[f'{name}/{wt}/{k}'] for k in ('c','dcda','dcdb','dcdr','dcdh')]
        dc=np.abs(r[0]-ref[0])/ref[0]
        print(name,wt,'c max rel %.2e nexact %d/%d'%(dc.max(),(dc==0).sum(),len(dc)),' kern',' '.join('%.1e'%rel(a,b) for a,b in zip(r[1:5],ref[1:])))
