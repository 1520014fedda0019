import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0,os.path.join(ROOT,'tests')); sys.path.insert(0,os.path.join(ROOT,'dr-using-scv-od_amd','pyshim'))
import numpy as np, torch, scvod_py, oracle_py, synth
P=scvod_py.make_params('semantickitti')
pts,_,_=synth.make_scan(5,0,'K64'); x=pts.numpy()
n=int(sys.argv[1])
x=x[:n]
ctx=scvod_py.Ctx(P,max_points_total=x.shape[0]+64,max_scans=1)
r=ctx.bin_scan(x,True,True); print('vox ok', n, r['n_apri'], r['n_voxels'], flush=True)
o=oracle_py.load()
b=o.bin(P,x,True); v=o.voxelize(P,b['apri'])
print('keys eq', np.array_equal(r['vox_key'],v['vox_key']), 'pts eq', np.array_equal(r['vox_pts'],v['vox_pts']), 'av eq', np.array_equal(r['vox_av'].view(np.uint32),v['vox_av'].view(np.uint32)),'cov eq', np.array_equal(r['vox_cov'].view(np.uint32),v['vox_cov'].view(np.uint32)))
cnt=np.diff(v['vox_pt_begin']); print('max voxel pts', cnt.max())
