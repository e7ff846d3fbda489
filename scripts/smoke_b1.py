import sys, time; sys.path.insert(0,'.')
import numpy as np
from oracle import oracle as O
O.build(ref=False)
from rfsurfhmc_amd.model.lib import libsurf, librf
rel=lambda a,b: np.abs(a-b).max()/max(np.abs(b).max(),1e-300)
thk=np.array([6.,6,13.,5,10,30,0]); vs=np.array([3.2,2.8,3.46,3.3,3.9,4.5,4.7])
vp,rho,_,_=O.empirical_relation(vs); t=np.arange(5.,41.)
c,f=libsurf.forward(thk,vp,vs,rho,t,'Rc'); co,fo=O.libsurf.forward(thk,vp,vs,rho,t,'Rc')
print('Rc fwd',f,fo,np.abs(c-co).max(), c[:3])
r=libsurf.adjoint_kernel(thk,vp,vs,rho,t,'Rc'); ro=O.libsurf.adjoint_kernel(thk,vp,vs,rho,t,'Rc')
print('Rc kern',[rel(a,b) for a,b in zip(r[:5],ro[:5])])
r=libsurf.adjoint_kernel(thk,vp,vs,rho,t,'Rg'); ro=O.libsurf.adjoint_kernel(thk,vp,vs,rho,t,'Rg')
print('Rg kern',[rel(a,b) for a,b in zip(r[:5],ro[:5])])
q=np.full(7,9999.)
a=(thk,rho,vp,vs,q,q,0.045,125,0.4,1.5,5.0,'freq',0.001,'P')
rf=librf.forward(*a); rfo=O.librf.forward(*a); print('rf fwd',rel(rf,rfo))
rf,kl=librf.kernel_all(*a); rfo,klo=O.librf.kernel_all(*a); print('rf kall',rel(rf,rfo),rel(kl,klo),[rel(kl[i],klo[i]) for i in range(4)])
