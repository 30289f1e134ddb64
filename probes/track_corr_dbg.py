import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from iggt_official_amd.heads.track_modules.blocks import CorrBlock
from oracle import restate_track
S, H, W, N = 2, 70, 91, 6
torch.manual_seed(5)
fm = torch.randn(S, 128, H, W)
targets = torch.randn(S, N, 128)
coords = torch.tensor([[10.0, 12.0], [10.5, 12.0], [10.0, 12.25], [33.3, 44.7], [0.0, 0.0], [89.5, 68.5]])[None].repeat(S, 1, 1)
ref = restate_track.corr_sample(restate_track.corr_pyramid(fm, 7), targets, coords, 4)
cb = CorrBlock(fm.permute(0, 2, 3, 1).contiguous().cuda(), num_levels=7, radius=4)
got = cb.corr_sample(targets.permute(1, 0, 2).contiguous().cuda(), coords.permute(1, 0, 2).contiguous().cuda())
got = got[:, :567].view(N, S, 567).permute(1, 0, 2).cpu()
for l in range(7):
    for n in range(N):
        d = (got[:, n, l * 81:(l + 1) * 81] - ref[:, n, l * 81:(l + 1) * 81]).abs().max()
        print(l, n, float(d), end=" | ")
    print()
print(got[0, 0, :12]); print(ref[0, 0, :12])
print(got[0, 0, :81].view(9, 9)[:3, :3]); print(ref[0, 0, :81].view(9, 9)[:3, :3])
