import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from align3r_amd import ops, _lib
def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).cuda()
def step(name, fn):
    print(">>", name, flush=True)
    r = fn()
    torch.cuda.synchronize()
    print("   ok", flush=True)
    return r
M, N, K = 200, 192, 96
x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3)
word = ops.absmax_word(x.device)
x2 = step("split+absmax", lambda: ops.split_fh2(x, 1.0, absmax=word)); print(ops.absmax_value(word), float(x.abs().max()))
w2 = step("split_w", lambda: ops.split_fh2_w(w))
y = step("linear plain", lambda: ops.linear_fh2(x2, w2, b))
print(float((y - ops.linear(x, w, b)).abs().max()))
word = ops.absmax_word(x.device)
y2 = step("linear out_fh2 + absmax", lambda: ops.linear_fh2(x2, w2, b, out_fh2=True, out_scale=32.0, out_absmax=word))
print(ops.absmax_value(word) / 32, float(y.abs().max()))
word = ops.absmax_word(x.device)
ln = step("layernorm absmax", lambda: ops.layernorm_fh2(rnd(333, 768, seed=5), rnd(768, seed=6), rnd(768, seed=7), absmax=word)); print(ops.absmax_value(word))
