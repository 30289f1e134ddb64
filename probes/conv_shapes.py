"""Per-shape time of the NHWC conv / resize calls in one 518^2 forward (geometry heads):
python probes/conv_shapes.py [S]   -> table sorted by total time, with useful TFLOP/s per shape."""
import os, sys, collections
os.environ.setdefault("MIOPEN_FIND_MODE", "2")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iggt_official_amd import _C
from iggt.models.vggt import IGGT

S = int(sys.argv[1]) if len(sys.argv) > 1 else 8
records = []
orig_conv, orig_rs = _C.conv2d_nhwc, _C.bilinear_ac_nhwc


def conv(x, w_hi, w_lo, bias, y, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = orig_conv(x, w_hi, w_lo, bias, y, **kw)
    e1.record()
    N, Hi, Wi, ldx = x.shape
    Cin = kw.get("Cin") or ldx
    Cout = kw.get("Cout") or w_hi.shape[0]
    Ho = kw.get("Ho") or y.shape[1]
    Wo = kw.get("Wo") or y.shape[2]
    key = ("conv", N, Hi, Wi, Cin, Cout, kw["KH"], kw.get("stride", 1), Ho, Wo, kw.get("prec", 3), kw.get("ps", 1),
           int(kw.get("res") is not None), int(bool(kw.get("relu_in"))))
    flops = 2.0 * N * Ho * Wo * Cout * Cin * kw["KH"] * kw["KW"]
    records.append((key, e0, e1, flops))
    return r


def rs(x, y, *a, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = orig_rs(x, y, *a, **kw)
    e1.record()
    records.append((("resize",) + tuple(x.shape) + tuple(y.shape[1:3]), e0, e1, 0.0))
    return r


_C.conv2d_nhwc, _C.bilinear_ac_nhwc = conv, rs
import iggt_official_amd.heads.convops as co
for mod in list(sys.modules.values()):
    if mod and getattr(mod, "__name__", "").startswith("iggt_official_amd"):
        if getattr(mod, "_C", None) is _C:
            pass
torch.manual_seed(0)
with torch.device("cuda"):
    model = IGGT(part_on_invalid_grid="skip").eval()
img = torch.rand(S, 3, 518, 518, device="cuda")
with torch.no_grad():
    model(img)
    records.clear()
    model(img)
torch.cuda.synchronize()
agg = collections.OrderedDict()
for key, e0, e1, fl in records:
    t = e0.elapsed_time(e1)
    a = agg.setdefault(key, [0, 0.0, 0.0])
    a[0] += 1; a[1] += t; a[2] += fl
tot = sum(a[1] for a in agg.values())
print(f"S={S}: {len(records)} calls, {tot:.2f} ms total")
for key, (n, t, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{t:8.3f} ms  x{n:3d}  {fl / t / 1e9 if t else 0:7.1f} TF/s  {key}")
