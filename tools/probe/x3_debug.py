import torch, torch.nn as nn, torch.nn.functional as F, os
from msml_amd import functional as Fh
torch.manual_seed(0)
n, h, w, c = 1, 14, 14, 64
x = torch.randn(n, h, w, c).cuda()
conv = nn.Conv2d(c, c, 3, 1, 1, bias=False).cuda()
xs = Fh.x3_from_f32(x)
got = Fh.x3_to_f32(Fh.conv_x3(xs, None, conv, None, None, None, None, False))
os.environ["MSML_NO_S2R_X3"] = "1"
gen = Fh.x3_to_f32(Fh.conv_x3(xs, None, conv, None, None, None, None, False))
torch.set_printoptions(precision=4, linewidth=200)
print("got", got[0, 0, 0, :16])
print("gen", gen[0, 0, 0, :16])
print("got", got[0, 3, 5, :16])
print("gen", gen[0, 3, 5, :16])
d = (got - gen).abs()
print("max diff per channel", d.amax((0, 1, 2)))
print("max diff per row", d.amax((0, 2, 3)))
print("max diff per col", d.amax((0, 1, 3)))
