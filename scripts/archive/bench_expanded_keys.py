import ctypes as C, sys, os, torch
sys.path.insert(0,'/root/repo')
from dilithium_amd import api, lib as dlib
api.init(0); L=dlib.load(); P=lambda t: C.c_void_p(t.data_ptr()); st=C.c_void_p(torch.cuda.current_stream().cuda_stream)
g=torch.Generator(device="cuda").manual_seed(5)
u8=lambda *sh: torch.randint(0,256,sh,dtype=torch.uint8,device="cuda",generator=g)
for level in (3,5,2):
    n=8192
    pk,sk=api.keygen(u8(n,32),level); mu=u8(n,64); sig,_=api.sign(sk,mu,level)
    A=api.expand_a(pk[:,:32].contiguous(),level); T=api.expand_t1(pk,level); vd=torch.empty(n,dtype=torch.int32,device="cuda")
    def t_us(fn,reps=60):
        for _ in range(10): fn()
        torch.cuda.synchronize(); e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        res=[]
        for r in range(3):
            e0.record()
            for _ in range(reps): fn()
            e1.record(); torch.cuda.synchronize(); res.append(e0.elapsed_time(e1)/reps*1e3)
        return sorted(res)[1]
    a=t_us(lambda: L.dil_verify_sig_expanded_dev(P(vd),P(A),P(pk),P(sig),P(mu),level,n,0,st))
    assert int(vd.abs().sum())==0
    b=t_us(lambda: L.dil_verify_sig_expanded2_dev(P(vd),P(A),P(T),P(pk),P(sig),P(mu),level,n,0,st))
    assert int(vd.abs().sum())==0
    w1p=torch.empty((n,{2:4,3:6,5:8}[level]*(192 if level==2 else 128)),dtype=torch.uint8,device="cuda")
    c=t_us(lambda: L.dil_verify_wire_core_dev(P(w1p),P(vd),P(A),P(pk),P(sig),level,n,0,st))
    print(f"level {level} n={n}: expanded A {a:.1f} us ({n/a:.1f} M/s)   expanded A + t1^ {b:.1f} us ({n/b:.1f} M/s)   [SIB + fused kernel alone {c:.1f} us]")
