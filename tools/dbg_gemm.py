"""torch-only repro: F.linear on a [3, 2100, 768] bf16 input that is a slice of [3, 2118, 768] (mode 'view') or its contiguous copy ('copy')"""
import sys, torch, torch.nn.functional as F
torch.manual_seed(0)
x = torch.randn(3, 2118, 768, device="cuda", dtype=torch.bfloat16)
W = torch.randn(1536, 768, device="cuda", dtype=torch.bfloat16) / 28
xs = x[:, :2100]
if sys.argv[1] == "copy":
    xs = xs.contiguous()
y = F.linear(xs, W)
torch.cuda.synchronize()
ref = (xs.float().reshape(-1, 768) @ W.float().t()).reshape(3, 2100, 1536)
print(sys.argv[1], "ok, contiguous out:", y.is_contiguous(), "max err", (y.float() - ref).abs().max().item(), flush=True)
