import ctypes as C, sys, torch
sys.path.insert(0,'/root/repo')
from dilithium_amd import api, lib as dlib
api.init(0); L=dlib.load(); P=lambda t: C.c_void_p(t.data_ptr()); st=C.c_void_p(torch.cuda.current_stream().cuda_stream)
g=torch.Generator(device="cuda").manual_seed(5)
def t_us(fn,reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    r=[]
    for _ in range(3):
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize(); r.append(e0.elapsed_time(e1)/reps*1e3)
    return sorted(r)[1]
for n in (1024,2048,4096,8192,16384,32768):
    ct=torch.randint(0,256,(n,32),dtype=torch.uint8,device="cuda",generator=g)
    sig=torch.randint(0,256,(n,3293),dtype=torch.uint8,device="cuda",generator=g)
    c=torch.empty((n,256),dtype=torch.int32,device="cuda")
    out=[]
    for cm in (1<<30,0):
        api.set_option("coop_max",cm)
        a=t_us(lambda: L.dil_sample_in_ball_dev(P(c),P(ct),3,n,st))
        out.append(a)
    print(f"SampleInBall (poly out) n={n:6d}: coop {out[0]:7.1f} us   lane {out[1]:7.1f} us")
