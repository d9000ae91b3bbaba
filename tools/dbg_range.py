import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import make_view_arrays
from align3r_amd.weights import TINY, synthetic_state_dict
from align3r_amd.engine import PairEngine
from align3r_amd import ops, _lib

sd = {k: np.array(v, copy=True) for k, v in synthetic_state_dict(TINY, 0).items()}
def mul(keys, f):
    for k in keys:
        sd[k] = (sd[k] * np.float32(f)).astype(np.float32)
heads = ("downstream_head1.dpt.", "downstream_head2.dpt.")
mul(("decoder_embed.weight", "decoder_embed.bias", "patch_embed_point_cloud.proj.weight", "patch_embed_point_cloud.proj.bias"), 2.0 ** 20)
mul([h + f"act_postprocess.{i}.0.weight" for h in heads for i in (1, 2)], 2.0 ** -20)
eng = PairEngine(TINY, sd)
eng.range_check = False
v = make_view_arrays(2, 64, 96)
args = [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in (v[0][0], v[1][0], v[0][1], v[1][1])]
out = eng.forward(*args)
st = eng.site_stats()
print("sites", len(st), "max", st.max(), "n>65504", int((st > 65504).sum()), "n nonfinite", int((~np.isfinite(st)).sum()))
print("idx >65504:", np.nonzero(st > 65504)[0][:40], st[st > 65504][:40])
for k, t in out.items():
    print(k, "finite" if bool(torch.isfinite(t).all()) else "NONFINITE", float(t.abs().max()))
for name in ("feat", "hook_a", "hook_b", "dec_last"):
    t = eng.tap(name, TINY.dec_embed_dim if name != "feat" else TINY.enc_embed_dim)
    print(name, "finite" if bool(torch.isfinite(t).all()) else "NONFINITE", float(t.abs().max()))

# resid_aux op check
def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).cuda()
M, N, K = 200, 192, 96
x, w, b, r = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3), rnd(M, N, seed=4)
w2 = ops.split_fh2_w(w)
x1, xs = ops.split_fh2(x), ops.split_fh2(x, 2.0 ** 6)
a1 = ops.Fh2(torch.zeros(M * N * 4, dtype=torch.uint8, device="cuda"), M, N)
a2 = ops.Fh2(torch.zeros(M * N * 4, dtype=torch.uint8, device="cuda"), M, N, 2.0 ** 5)
y1 = ops.linear_fh2(x1, w2, b, epi=_lib.EPI_RESID, resid=r, aux_fh2=a1, aux_relu=True)
y2 = ops.linear_fh2(xs, w2, b, epi=_lib.EPI_RESID, resid=r, aux_fh2=a2, aux_relu=True, out_scale=2.0 ** 5)
ref = torch.relu(y1).double()
print("a1 err", float((a1.value() - ref).abs().max()), "a2 err", float((a2.value() - ref).abs().max()), "y diff", float((y1 - y2).abs().max()))
print("a2 planes max", float(a2.planes().abs().max()), "a1 planes max", float(a1.planes().abs().max()))
