import os, sys, torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_unet_kernels as T
from mm_masking_amd import unet_hip as uh
DEV = torch.device("cuda:0")
H = 64
model = T._policy(0.0, torch.float32); model.train()
g = torch.Generator().manual_seed(1)
x = torch.rand(2, 1, H, H, generator=g).to(DEV); gsel = torch.randn(2, H, H, generator=g).to(DEV)
q = T._Q.apply
keep = {}
def conv(t, m, name):
    y = q(F.relu(F.conv2d(t, m.weight.to(torch.bfloat16).float(), m.bias, padding=1))); y.retain_grad(); keep[name] = y; return y
t = [None] * 6
xb = x.to(torch.bfloat16).float()
cur = conv(conv(xb, model.encoder[0][0], "a_e0"), model.encoder[0][2], "d_e0"); t[0] = cur
for i in range(1, 6):
    cur = conv(conv(t[i - 1], model.encoder[i][0], "a_e%d" % i), model.encoder[i][2], "d_e%d" % i)
    t[i] = F.max_pool2d(cur, 2, 2); t[i].retain_grad(); keep["t_%d" % i] = t[i]
cur = t[5]
for j in range(5):
    skip = t[4 - j]
    u = q(F.interpolate(cur, size=skip.shape[2:], mode="bilinear", align_corners=True)); u.retain_grad(); keep["u_%d" % j] = u
    d1 = conv(conv(u, model.decoder[j][0], "a1_%d" % j), model.decoder[j][2], "d1_%d" % j)
    cur = conv(conv(torch.cat([skip, d1], 1), model.decoder[j][0], "a2_%d" % j), model.decoder[j][2], "d2_%d" % j)
fl = model.final_layer[0]
ref = torch.sigmoid(F.conv2d(cur, fl.weight.to(torch.bfloat16).float(), fl.bias)).squeeze(1)
(ref * gsel).sum().backward()
model.zero_grad()
uh.DEBUG = {}
out = uh.unet_mask(model, x, True, 0); (out * gsel).sum().backward()
D = uh.DEBUG
def cmp(name, mine, y, post_act=True):
    gy = y.grad
    want = gy.to(torch.bfloat16).float()
    if post_act: want = want * (y > 0)
    want = want.permute(0, 2, 3, 1)
    rel = ((mine.float() - want).norm() / (want.norm() + 1e-20)).item()
    print("%-12s rel err %.4f  |ref| %.4g" % (name, rel, want.norm().item()))
cmp("gz_d2_4", D["gz_d2_4"], keep["d2_4"])
for j in range(4, -1, -1):
    cmp("gz_a2_%d" % j, D["gz_a2_%d" % j], keep["a2_%d" % j])
    cmp("gz_d1_%d" % j, D["gz_d1_%d" % j], keep["d1_%d" % j])
    cmp("gz_a1_%d" % j, D["gz_a1_%d" % j], keep["a1_%d" % j])
    cmp("g_u_%d" % j, D["g_u_%d" % j], keep["u_%d" % j], post_act=False)
for i in range(5, 0, -1):
    cmp("g_t_%d" % i, D["g_t_%d" % i], keep["t_%d" % i], post_act=False)
    cmp("gz_d_e%d" % i, D["gz_d_e%d" % i], keep["d_e%d" % i])
