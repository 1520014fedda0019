import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0,os.path.join(ROOT,'tests')); sys.path.insert(0,os.path.join(ROOT,'dr-using-scv-od_amd','pyshim'))
import numpy as np, torch, scvod_py, oracle_py, synth
o=oracle_py.load()
P=scvod_py.make_params('parkinglot')
rng = np.random.default_rng(3)
x = rng.uniform(-45, 45, (30000, 4)).astype(np.float32)
x[:, 2] = rng.uniform(-3, 6, 30000)
x[:100, 1] = 0.0
x[100:110, :2] = 0.0
ctx=scvod_py.Ctx(P,max_points_total=200000,max_scans=1)
r=ctx.bin_scan(x,True,False); b=o.bin(P,x,True)
print('n', r['n_apri'], len(b['apri']))
for f in r['apri'].dtype.names:
    a=r['apri'][f]; c=b['apri'][f]
    if a.dtype==np.float32: bad=np.nonzero(a.view(np.uint32)!=c.view(np.uint32))[0]
    else: bad=np.nonzero(a!=c)[0]
    print(f, len(bad), [(a[i],c[i]) for i in bad[:3]])
    if len(bad): 
        i=bad[0]; print('   pt', r['apri'][i], b['apri'][i])
P=scvod_py.make_params('semantickitti')
ctx2=scvod_py.Ctx(P,max_points_total=200000,max_scans=1)
pts,_,_=synth.make_scan(5,0,'K64'); x=pts.numpy()
r=ctx2.patchwork(x); oo=o.patchwork(P,x,1)
print('cls eq', np.array_equal(r['cls'],oo['cls']), (r['cls']!=oo['cls']).sum())
print('g eq', np.array_equal(r['ground_idx'],oo['ground_idx']), 'ng eq', np.array_equal(r['nonground_idx'],oo['nonground_idx']))
for f in ('n_pts','n_ground','status'):
    bad=np.nonzero(r['planes'][f]!=oo['planes'][f])[0]; print(f, len(bad), bad[:5], r['planes'][f][bad[:5]], oo['planes'][f][bad[:5]])
for f in ('normal','mean','sv'):
    bad=np.nonzero((r['planes'][f].view(np.uint32)!=oo['planes'][f].view(np.uint32)).any(1) & (oo['planes']['status']>0))[0]
    print(f, len(bad), bad[:5]); 
    if len(bad): print(r['planes'][f][bad[0]], oo['planes'][f][bad[0]], oo['planes']['n_pts'][bad[0]])
