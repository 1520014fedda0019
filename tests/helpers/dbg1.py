import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0,os.path.join(ROOT,'tests')); sys.path.insert(0,os.path.join(ROOT,'dr-using-scv-od_amd','pyshim'))
import numpy as np, torch, scvod_py, oracle_py, synth
P=scvod_py.make_params('semantickitti')
print(torch.cuda.get_device_name(0), flush=True)
pts,_,_=synth.make_scan(5,0,'K64'); x=pts.numpy()
ctx=scvod_py.Ctx(P,max_points_total=x.shape[0]+64,max_scans=1)
print('arena MB', ctx.arena_bytes()/1e6, flush=True)
mode=sys.argv[1] if len(sys.argv)>1 else 'bin'
if mode=='bin':
    r=ctx.bin_scan(x,True,False); print('bin ok', r['n_apri'], flush=True)
    r=ctx.bin_scan(x,True,True); print('vox ok', r['n_voxels'], flush=True)
elif mode=='pw':
    r=ctx.patchwork(x); print('pw ok', r['n_ground'], r['n_nonground'], flush=True)
else:
    r=ctx.process_scan(x); print('all ok', r['n_voxels'], flush=True)
