import sys, time; sys.path.insert(0,'.'); sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import torch
from raytracer_amd import *
W,H,spp=1920,1080,256
b=SceneBuilder.sample("Random Spheres", CameraConfiguration(W,H))
sc=Scene(b,0); st=RenderSettings(W,H,8,spp)
stream=torch.cuda.current_stream().cuda_stream
for world in (1,2,4,8):
    rows=strip_rows(H,8,0,world)
    buf=torch.zeros((rows,W,3),dtype=torch.float32,device="cuda:0")
    ts=[]
    for r in range(min(world,3)):
        for it in range(3):
            torch.cuda.synchronize(); t0=time.perf_counter()
            sc.render_strips_device(st,8,r,world,12345,buf.data_ptr(),None,None,stream)
            torch.cuda.synchronize(); ts.append(time.perf_counter()-t0)
    if world == 1: one = min(ts)
    print(world, "rank-time ms min/max", round(min(ts)*1e3,2), round(max(ts)*1e3,2), "ideal", round(one*1e3/world,2))
