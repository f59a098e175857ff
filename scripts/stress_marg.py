import sys
sys.path.insert(0,'ground-fusion_amd'); sys.path.insert(0,'oracle')
import numpy as np, gfamd, oracle_py as O, synth_window as SW
est=gfamd.Estimator()
w=SW.make_window(4,O); O.ba_solve(w,4)
po=O.ba_marginalize(w,0)
def inv(p):
    n=p['n']; J=p['J'].reshape(n,n); return J.T@J, J.T@p['r']
Ao,bo=inv(po)
worst=0
for it in range(40):
    pg=est.marginalize([w],0)[0]
    Ag,bg=inv(pg)
    sc=np.sqrt(np.outer(np.diag(Ao),np.diag(Ao)))+1e-6*np.abs(Ao).max()
    e=(np.abs(Ao-Ag)/sc).max()
    worst=max(worst,e)
    if e>1e-6:
        i,j=np.unravel_index(np.argmax(np.abs(Ao-Ag)/sc),Ao.shape)
        print('iter',it,'err',e,'at',i,j,Ao[i,j],Ag[i,j], 'n',pg['n'],'m',pg['m'])
print('worst',worst)
