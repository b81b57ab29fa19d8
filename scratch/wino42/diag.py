"""Phase stamps (s_memtime, relative to kernel entry) of the last workgroup of wino42_kernel from a -DWINO42_EXP=128 build
(scratch/wino42/build_ablations.sh 128; SRGAN_HIP_LIB=scratch/wino42/lib_128.so).  Columns per wave: loop start, loop end,
X image stored, barrier passed, output transform's column pass done, kernel end."""
import sys, os
_R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "style-restricted_gan_amd"))
import torch
from srgan_amd import ops
B = int(os.environ.get("B", "32"))
for name, ci, h, co in (("G.down1", 64, 128, 128), ("G.down2", 128, 64, 256)):
    x = torch.randn(B, h, h, ci, device="cuda").permute(0, 3, 1, 2)
    w = torch.randn(co, ci, 4, 4, device="cuda") / (ci * 16) ** 0.5
    for rep in range(3):
        y = ops.conv2d(x, w, None, 2, 1)
        torch.cuda.synchronize()
    v = y.permute(0, 2, 3, 1).reshape(-1)[:96].view(8, 12).cpu()
    print(name, "fwd: wave nk | loop_start loop_end x_stored barrier colpass end | item_top transformed loads_issued")
    for wv in range(8):
        t = v[wv].tolist()
        print(f"  {wv} {t[0]:4.0f} | " + " ".join(f"{a:9.0f}" for a in t[1:10]))
    gy = torch.randn_like(y)
    desc = ops._conv_desc(B, h, h, ci, h // 2, h // 2, co, 4, 4, 2, 1, 0, w)
    dx = torch.empty_like(x)
    for rep in range(3):
        ops._run_conv_dgrad(desc, gy, w, dx)
        torch.cuda.synchronize()
    v = dx.permute(0, 2, 3, 1).reshape(-1)[:96].view(8, 12).cpu()
    print(name, "dgrad: wave nk | loop_start loop_end x_stored barrier colpass end | item_top transformed loads_issued")
    for wv in range(8):
        t = v[wv].tolist()
        print(f"  {wv} {t[0]:4.0f} | " + " ".join(f"{a:9.0f}" for a in t[1:10]))
