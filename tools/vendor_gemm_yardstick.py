import torch
torch.backends.cuda.matmul.allow_tf32 = False
dev="cuda"
shapes=[("lin2_fwd",8192,512,4096),("lin1_fwd",8192,4096,512),("qkv_fwd",8192,1536,512),("dec_fwd",8192,33000,512),("dec_dgrad",8192,512,33000),("lin_wgrad(TN)",4096,512,8192)]
for name,m,n,k in shapes:
    if "TN" in name:
        a=torch.randn(k,m,device=dev); b=torch.randn(k,n,device=dev); f=lambda: a.t()@b
    elif "dgrad" in name:
        a=torch.randn(m,k,device=dev); b=torch.randn(k,n,device=dev); f=lambda: a@b
    else:
        a=torch.randn(m,k,device=dev); b=torch.randn(n,k,device=dev); f=lambda: a@b.t()
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/20
    print(f"{name:16s} {m}x{n}x{k}: {ms:.3f} ms {2*m*n*k/ms/1e9:.1f} TFLOP/s (torch.matmul fp32, vendor library)")
