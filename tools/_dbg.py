import sys; sys.path.insert(0,'/root/repo')
import numpy as np
from rgbd360_amd import synth
from rgbd360_amd.register import RegisterPhotoICP
from oracle import oracle as O
(rgbA,dA),(rgbB,dB),T=synth.make_pair(256,128,seed=1234)
reg=RegisterPhotoICP(); reg.setNumPyr(3); reg.setTargetFrame(rgbA,dA); reg.setSourceFrame(rgbB,dB)
ora=O.Oracle(n_pyr=3,math_mode=1,reduce_mode=1); ora.set_target(rgbA,dA); ora.set_source(rgbB,dB)
e=reg.eval(0,np.eye(4),2)
rms,err2,nv=ora.error(0,np.eye(4),2)
H,g,Hd,gd,nvis=ora.hessgrad(0,np.eye(4),2)
print('err2',e['err2'],err2,'split',e['err2_split'],'nvalid',e['n_valid'],nv,'nvis',e['n_visible'],nvis)
k=0; 
tot_gpu=[]; tot_ref=[]
for a in range(6):
    for b in range(a,6):
        tot_gpu.append(e['H64'][a,b]); tot_ref.append(Hd[a,b])
tot_gpu+=list(e['g64']); tot_ref+=list(gd)
for i,(x,y) in enumerate(zip(tot_gpu,tot_ref)): print(i,'%.6g %.6g'%(x,y), 'OK' if abs(x-y)<=2e-5*abs(Hd).max() else '<<<')
