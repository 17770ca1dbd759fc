import sys, torch
sys.path.insert(0, "/root/repo")
import detectron2_centernet_amd as pkg
from detectron2_centernet_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
cin, cout, H = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
x = torch.randn(64, H, H, cin, generator=g).half().to(dev)
w = (torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5).to(dev)
p = ops.PackedConv(w, None, None, stride=1, pad=1, compute=ops.F16)
for _ in range(3):
    y = ops.conv2d(x, p, act=ops.ACT_RELU)
torch.cuda.synchronize()
t = y.view(-1)[:160].view(torch.int64).cpu().tolist()
print("ts", t)
print("d ", [t[i+1]-t[i] for i in range(len(t)-1)])
