import sys, os, time, torch
sys.path.insert(0, "/root/repo")
from chatts_amd import _lib
import ctypes as C
lib=_lib.load(); DEV="cuda"
T, nq, nkv, d, ctx = 798, 40, 8, 128, 1024
qkv=torch.randn((T,(nq+2*nkv)*d),device=DEV)
kc=torch.randn((nkv,ctx,d),device=DEV); vc=torch.randn((nkv,ctx,d),device=DEV)
out=torch.empty((T,nq*d),device=DEV)
cache=_lib.KvCache(k=kc.data_ptr(),v=vc.data_ptr(),max_ctx=ctx)
st=torch.cuda.current_stream()
NS=int(sys.argv[1]) if len(sys.argv)>1 else 1
wsb=int(lib.chatts_attn_workspace(T,nq,NS)); ws=torch.empty(max(wsb,16),dtype=torch.uint8,device=DEV)
def run():
    _lib.check(lib.chatts_attention(qkv.data_ptr(),T,nq,nkv,0,None,C.byref(cache),out.data_ptr(),NS,ws.data_ptr(),wsb,st.cuda_stream))
run(); torch.cuda.synchronize()
e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
e0.record(st)
for _ in range(20): run()
e1.record(st); torch.cuda.synchronize()
print("attention prefill T=798 key-splits %d: %.1f us" % (NS, e0.elapsed_time(e1)*1e3/20))
