import os, sys, torch
sys.path.insert(0, "intro-tc-vae_amd")
from hipvae import functional as HF
d = torch.device("cuda:0")
for Ci, Co, S in ((128, 128, 32), (64, 64, 64), (256, 256, 16), (512, 512, 8)):
    B = 64
    x = torch.randn(B, Ci, S, S, device=d); dy = torch.randn(B, Co, S, S, device=d)
    xp, dyp = HF.split_planes(x, 2), HF.split_planes(dy, 2)
    nws = HF.lib.itcv_conv2d_wgrad_bf16p_workspace(B, Ci, S, S, Co, 3)
    ws = torch.zeros(nws, dtype=torch.uint8, device=d)
    dw = torch.empty(Co, Ci, 3, 3, device=d)
    for _ in range(10):
        HF.call("itcv_conv2d_wgrad_bf16p", HF.ptr(xp), HF.ptr(dyp), HF.ptr(dw), B, Ci, S, S, Co, 3, 0, 0, HF.ptr(ws), nws, HF.stream())
    torch.cuda.synchronize()
    v = ws.view(torch.float32)[:2].tolist()
    print(f"{Ci}->{Co}@{S}: loop {v[0]:.0f} cycles / {v[1]:.0f} steps = {v[0]/max(v[1],1):.0f} per step (MFMA ideal 2304)")
