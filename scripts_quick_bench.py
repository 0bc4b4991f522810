import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'multimodal-long-transformer-2021_amd'))
import torch, mmt_amd
torch.manual_seed(0)
def run(B,S,N,dtype,radius,g0,ng,iters=20):
    q,k,v = (torch.randn(B,S,N,64,device='cuda',dtype=dtype) for _ in range(3))
    emb = torch.randn(32,N,64,device='cuda',dtype=dtype)*0.5; bias = torch.randn(32,N,device='cuda',dtype=dtype)*0.5
    pat = mmt_amd.AttentionPattern(local_radius=radius, global_start=g0, n_global=ng, id_mode=1, max_dist=12)
    for _ in range(3): mmt_amd.relative_attention_forward(q,k,v,emb,bias,pattern=pat)
    torch.cuda.synchronize()
    e0,e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): mmt_amd.relative_attention_forward(q,k,v,emb,bias,pattern=pat)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)/iters
    byts = 4*B*S*N*64*q.element_size()
    print(f'B{B} S{S} N{N} {dtype} r{radius} g{ng}: {ms*1e3:.1f} us  algGB/s={byts/ms/1e6:.0f}', flush=True)
run(4,4096,12,torch.bfloat16,64,3971,8)
run(4,4096,12,torch.bfloat16,64,0,0)
run(8,1024,12,torch.float32,64,786,8)
run(1,4096,12,torch.bfloat16,1<<30,0,0,iters=5)
