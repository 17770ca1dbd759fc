import sys, os, torch, tempfile, pathlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_model_gpu import make_model, cpu_state_dict
from oracle import ctdet_oracle as O, model_ref as MR
from detectron2_centernet_amd.engine import train_step as TS
from detectron2_centernet_amd import ops
dev = torch.device("cuda:0")
model, cfg = make_model(pathlib.Path(tempfile.mkdtemp()), "f16", seed=11)
model.train()
sd = cpu_state_dict(model)
g = torch.Generator().manual_seed(0)
img = torch.randint(0, 256, (2, 3, 128, 128), generator=g, dtype=torch.uint8)
x_ref, _ = O.preprocess([i for i in img], cfg.MODEL.PIXEL_MEAN, cfg.MODEL.PIXEL_STD, 32)
x = ops.preprocess(img.to(dev), cfg.MODEL.PIXEL_MEAN, cfg.MODEL.PIXEL_STD, 128, 128)
n = MR.Net(sd, True)
with torch.no_grad():
    base_ref = MR.dla_base(n, "backbone.base", x_ref, (1, 1, 1, 2, 2, 1))
    base = TS.dla_base(model.backbone.base, x)
    for i, (a, b) in enumerate(zip(base, base_ref)):
        e = (a.float().cpu().permute(0, 3, 1, 2) - b).abs().max().item()
        print("base level", i, "max err", e, "ref max", b.abs().max().item())
    ups_ref = MR.dla_up(n, "backbone.dla_up", base_ref, 2)
    ups = TS.dla_up(model.backbone.dla_up, base)
    for i, (a, b) in enumerate(zip(ups, ups_ref)):
        e = (a.float().cpu().permute(0, 3, 1, 2) - b).abs().max().item()
        print("dla_up", i, "max err", e, "ref max", b.abs().max().item())
    y_ref = [u.clone() for u in ups_ref[:3]]
    MR.ida_up(n, "backbone.ida_up", y_ref, 0, 3)
    y = list(ups[:3]); TS.ida_up(model.backbone.ida_up, y, 0, 3)
    e = (y[-1].float().cpu().permute(0, 3, 1, 2) - y_ref[-1]).abs().max().item()
    print("ida_up out max err", e, "ref max", y_ref[-1].abs().max().item())
    z_ref = MR.centernet_heads(n, y_ref[-1])
    z = TS.heads(model, y[-1])
    for k in z:
        a = z[k].float().cpu().permute(0, 3, 1, 2)[:, :z_ref[k].shape[1]]
        print(k, "max err", (a - z_ref[k]).abs().max().item(), "ref max", z_ref[k].abs().max().item())
