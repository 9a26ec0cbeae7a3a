"""Analysis aid (CPU, oracle): traversal steps of every Scene.Hit call of every pixel of config 4, frame 1.
Shows what bounds the frame on the GPU: the serial chain of the heaviest pixel / 8x8 block."""
import sys, time, ctypes as C
from pathlib import Path; ROOT = Path(__file__).resolve().parents[1]; sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tests'))
import numpy as np
import oracle_binding as ob
from yetanotherconsolegameengine_amd import abi, scenes
from yetanotherconsolegameengine_amd.scene import flatten
sc,w,h,ss,pose=scenes.config_scene(4)
o=ob.OracleRenderer(sc,w,h,ss,pose,flat=flatten(sc))
o.render(stages=0,threads=8)
n=o.hiW*o.hiH
q=np.zeros((n,8),np.uint32)
o.L.orc_query_profile.restype=C.c_int; o.L.orc_query_profile.argtypes=[C.c_void_p,C.c_void_p,C.c_int]
t=time.time(); print(o.L.orc_query_profile(o.ctx,q.ctypes.data,8), time.time()-t)
tri=q>>16; q=q&0xffff
np.save('/tmp/ycge_query_profile.npy',q)
tot=q.sum(1).astype(np.int64)
print('per-pixel total steps: max',tot.max(),'p99.9',np.percentile(tot,99.9))
idx=np.argsort(-tot)[:12]
for i in idx: print(i%o.hiW,i//o.hiW,tot[i],q[i],'triangle tests',tri[i])
# ray-parallel chain: P + max(S1,S2, B + max(S1',S2'))
P=q[:,0]; S1=q[:,1]; S2=q[:,2]; B=q[:,3]; S1b=q[:,4]; S2b=q[:,5]
par=P+np.maximum(np.maximum(S1,S2),B+np.maximum(S1b,S2b))
print('ray-parallel chain max',par.max(),'vs serial',tot.max())
W,H=o.hiW,o.hiH
def blockmax(a): return a.reshape(H//8,8,W//8,8).max(axis=(1,3))
# megakernel wave model: sum over query slots of per-block max
mk=sum(blockmax(q[:,k].astype(np.int64)) for k in range(8))
print('wave model (sum over queries of block max): max',mk.max(), ' per-pixel-sum block max',blockmax(tot).max())

# what two triangles per step would give: a leaf of n triangles costs ceil(n/2) steps; bound by halving the triangle steps
q2=(q-tri)+(tri+1)//2
mk2=sum(blockmax(q2[:,k].astype(np.int64)) for k in range(8))
print('wave model with 2 triangles per step: max',mk2.max(),'(now',mk.max(),')')

# query fan-out: the queries of one hit (shadow rays + bounce) traced side by side, per 8x8 block
bm = lambda k: blockmax(q2[:, k].astype(np.int64))
fan = bm(0) + np.maximum(np.maximum(bm(1), bm(2)), bm(3)) + np.maximum(bm(4), bm(5)) + bm(6) + bm(7)
print('fan-out wave model (2 tris/step): max', fan.max(), 'vs', mk2.max())
for thr in (128, 192, 256, 384):
    sel = mk2 >= thr
    print(f'  blocks with wave model >= {thr}: {int(sel.sum())}; after fan-out their max {fan[sel].max()}, median ratio {np.median(fan[sel] / mk2[sel]):.2f}')
order = np.argsort(-mk2.ravel())[:10]
print('  heaviest blocks now -> fanned:', [(int(mk2.ravel()[i]), int(fan.ravel()[i])) for i in order])
