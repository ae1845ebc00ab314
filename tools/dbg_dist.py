import sys, torch
sys.path[:0] = ['.', 'bundle-adjusting-gaussian-splatting_amd', 'tests']
from parity import run_hip, run_oracle
from scenes import make_case
scene, cam = make_case(1500, 128, 96, 2.0, 1, seed=13)
outs, _, views = run_hip(scene, cam, 1, None, depth_key="distance")
st, _ = run_oracle(scene, cam, 1, None, depth_key="distance")
vis = st.pre.visible
d_o = st.pre.depth.detach().float()
d_h = views["depth_bits"].view(torch.float32)
bad = (d_o.view(torch.int32) != views["depth_bits"]) & vis
print("bad", int(bad.sum()), "of", int(vis.sum()))
idx = bad.nonzero().squeeze(1)[:8]
for i in idx:
    x, y, z = scene["means3D"][i].tolist()
    print(i.item(), d_o[i].item(), d_h[i].item(), (d_o[i].view(torch.int32) - views["depth_bits"][i]).item(), x, y, z)
    tx, ty, tz = torch.tensor(x), torch.tensor(y), torch.tensor(z) + 4.0
    print("  manual", torch.sqrt(tx*tx + ty*ty + tz*tz).item(), torch.sqrt((tx*tx + ty*ty + 1e-20) + tz*tz).item())
