import sys, os
sys.path.insert(0,'dr-using-scv-od_amd/pyshim')
import numpy as np, torch, scvod_py, synth
kind=sys.argv[2] if len(sys.argv)>2 else "K64"; preset=sys.argv[3] if len(sys.argv)>3 else "semantickitti"
P=scvod_py.make_params(preset)
n=int(sys.argv[1]) if len(sys.argv)>1 else 600; skip=5
parts=[];offs=[0];poses=[]
for i in range(n):
    p,l,pose=synth.make_scan(5,i,kind,device="cuda"); parts.append(p); offs.append(offs[-1]+p.shape[0]); poses.append(pose)
pts=torch.cat(parts).contiguous(); offs=np.asarray(offs,np.int32)
ctx=scvod_py.Ctx(P,max_points_total=int(offs[-1])+64,max_scans=n)
nxt=np.array([s+skip if s+skip<n else -1 for s in range(n)],np.int32)
T=np.zeros((n,12),np.float32)
for s in range(n-skip): T[s]=ctx.pose_delta(poses[s],poses[s+skip])
ctx.batch_process(pts,offs); ctx.batch_cluster(); ctx.batch_cluster_types()
for seg,warm in ((11,12),(0,12)):
    ctx.set_track_mode(True,seg,warm)
    ctx.batch_track(T,next_scan=nxt)
    ctx.set_timing(True)
    ctx.batch_track(T,next_scan=nxt)
    print(seg,warm,[ (k,round(v,3)) for k,v in ctx.timings() if 'chain' in k], ctx.batch_track_stats())
    ctx.set_timing(False)
