"""GPU: RAFT2 / SEA-RAFT optical flow (csrc/raft.hip through the C ABI a3r_raft_*; default arithmetic since round 3: the
two-plane fp16 kernels, with the three-plane bf16 ones as the range fallback) against goldens produced by the
reference's own third_party/RAFT modules in the build container with the build's synthetic weights (tests/golden/raft.npz,
tests/golden/make_goldens.py --only raft; inputs are rebuilt here by align3r_amd.raft_weights.synthetic_raft_frames).

  * TINY configuration, 2 pairs 128 x 160, 3 iterations: every stage -- context network + init_conv, both feature maps, the four
    correlation-pyramid levels, the first heads, the first correlation lookup and motion features, hidden state and coarse flow after
    every iteration -- at max|a - b| / max|b| < 1e-4, and the final up-sampled flow at 1e-4;
  * RAFT_M (the configuration the reference's load_RAFT builds), called as the reference calls it (iters = 20, test_mode = True):
    first prediction and final flow, both at 1e-4 (measured: 2-5e-6 after 20 recurrent steps).
"""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, record_margin, rel_err
from align3r_amd.raft_weights import RAFT_M, RAFT_TINY, synthetic_raft_frames, synthetic_raft_state_dict

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def g():
    return np.load(os.path.join(GOLDEN, "raft.npz"))


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_raft_tiny_stages_and_flow(g):
    from align3r_amd.raft import RaftEngine
    cfg = RAFT_TINY
    eng = RaftEngine(cfg, synthetic_raft_state_dict(cfg, 0))
    B, H, W, iters = 2, 128, 160, 3
    h, w, d = H // 8, W // 8, cfg.dim
    i1, i2 = synthetic_raft_frames(B, H, W, 7)
    ccp = (cfg.corr_channel + 31) // 32 * 32
    z = lambda *s: torch.zeros(*s, device="cuda")
    taps = dict(cnet=z(B, h, w, 2 * d), fmap=z(2 * B, h, w, 2 * d), flow_update0=z(B, h, w, 6), weight0=z(B, h, w, 576),
                lookup0=z(B, h, w, ccp), motion0=z(B, h, w, d))
    hl, wl = h, w
    for l in range(cfg.corr_levels):
        taps[f"corr_pyr{l}"] = z(B * h * w, hl, wl)
        hl, wl = hl // 2, wl // 2
    for it in range(iters):
        taps[f"net{it}"] = z(B, h, w, d)
        taps[f"flow8_{it}"] = z(B, h, w, 2)
    flow = eng.forward(dev(i1), dev(i2), iters=iters, taps=taps)
    t = {k: v.cpu().numpy() for k, v in taps.items()}
    m = {}
    m["cnet"] = rel_err(t["cnet"], g["t_cnet"])
    m["fmap1"] = rel_err(t["fmap"][:B], g["t_fmap1"])
    m["fmap2"] = rel_err(t["fmap"][B:], g["t_fmap2"])
    for l in range(cfg.corr_levels):
        m[f"corr_pyr{l}"] = rel_err(t[f"corr_pyr{l}"], g[f"t_corr_pyr{l}"][:, 0])
    m["flow_update0"] = rel_err(t["flow_update0"], g["t_flow_update0"])
    m["weight0"] = rel_err(t["weight0"], g["t_weight0"])
    m["lookup0"] = rel_err(t["lookup0"][..., :cfg.corr_channel], g["t_corr_lookup0"])
    assert not t["lookup0"][..., cfg.corr_channel:].any()           # the K padding of convc1 is zero-filled
    m["motion0"] = rel_err(t["motion0"], g["t_motion0"])
    for it in range(iters):
        m[f"net{it}"] = rel_err(t[f"net{it}"], g[f"t_net_{it}"])
        m[f"flow8_{it}"] = rel_err(t[f"flow8_{it}"], g[f"t_flow8_{it}"])
    m["flow"] = rel_err(flow.cpu().numpy(), g["t_flow"])
    record_margin("raft_tiny_vs_reference", **m)
    bad = {k: v for k, v in m.items() if not v < TOL}
    assert not bad, bad
    # zero iterations: the prediction from the context network alone (raft.py:213-220)
    f0 = eng.forward(dev(i1), dev(i2), iters=0)
    assert rel_err(f0.cpu().numpy(), g["t_flow_up_0"]) < TOL
    # batch invariance: pair 1 alone
    f1 = eng.forward(dev(i1[1:]), dev(i2[1:]), iters=iters)
    assert rel_err(f1.cpu().numpy(), flow[1:].cpu().numpy()) < 1e-5


@pytest.mark.parametrize("tag,H,W,seed", [("m1", 128, 160, 11), ("m2", 160, 192, 13)])
def test_raft_m_as_the_reference_calls_it(g, tag, H, W, seed):
    """RAFT_M, iters = 20: the first iteration's prediction and the final flow (|flow| up to ~130 px with these random weights, 20
    recurrent steps) at 1e-4 of their maximum (measured: 2-5e-6; margins are recorded)."""
    from align3r_amd.raft import RAFT2
    net = RAFT2(RAFT_M, synthetic_raft_state_dict(RAFT_M, 0)).to("cuda").eval()
    i1, i2 = synthetic_raft_frames(1, H, W, seed)
    one = net(dev(i1), dev(i2), iters=1, test_mode=True)[1].cpu().numpy()
    out = net(dev(i1), dev(i2), iters=20, test_mode=True)
    assert isinstance(out, list) and len(out) == 2 and tuple(out[1].shape) == (1, 2, H, W)
    e1, e20 = rel_err(one, g[f"{tag}_flow_up_1"]), rel_err(out[1].cpu().numpy(), g[f"{tag}_flow"])
    record_margin(f"raft_m_{tag}_vs_reference", flow_iter1=e1, flow_iter20=e20)
    assert e1 < TOL, e1
    assert e20 < TOL, e20


def test_raft_per_frame_feature_cache_is_bitwise_the_full_forward():
    """a3r_raft_encode + a3r_raft_forward_features against a3r_raft_forward: a frame's feature map does not depend on the batch it was
    encoded in nor on its partner, so the flow from cached per-frame features equals the full forward's bit for bit (what
    cloud_opt_flow.get_flow relies on when it encodes every frame once)."""
    from align3r_amd.raft import RaftEngine
    eng = RaftEngine(RAFT_TINY, synthetic_raft_state_dict(RAFT_TINY, 0))
    i1, i2 = synthetic_raft_frames(3, 128, 160, 17)
    frames = dev(np.concatenate([i1, i2]))                         # six frames
    fm_all = eng.encode(frames)                                    # one batch of six
    fm_one = torch.cat([eng.encode(frames[k:k + 1].contiguous()) for k in range(6)])
    assert torch.equal(fm_all, fm_one)                             # batch-invariant
    full = eng.forward(frames[:3].contiguous(), frames[3:].contiguous(), iters=4)
    cached = eng.forward(frames[:3].contiguous(), frames[3:].contiguous(), iters=4, fmaps=(fm_all[:3].contiguous(), fm_all[3:].contiguous()))
    assert torch.equal(full, cached)
    back = eng.forward(frames[3:].contiguous(), frames[:3].contiguous(), iters=4, fmaps=(fm_all[3:].contiguous(), fm_all[:3].contiguous()))
    assert torch.equal(back, eng.forward(frames[3:].contiguous(), frames[:3].contiguous(), iters=4))
    one = eng.forward(frames[1:2].contiguous(), frames[4:5].contiguous(), iters=4, fmaps=(fm_all[1:2].contiguous(), fm_all[4:5].contiguous()))
    assert torch.equal(one, cached[1:2])                           # a pair's flow does not depend on the batch it is computed in
    with pytest.raises(RuntimeError, match="fmap1"):
        eng.forward(frames[:3].contiguous(), frames[3:].contiguous(), fmaps=(fm_all[:2].contiguous(), fm_all[3:].contiguous()))


def test_raft_fp16_range_fallback(monkeypatch):
    """The default arithmetic stores activations as two fp16 planes with scale 1: a call whose activations reach 2^15 is detected by
    the range statistic (a3r_raft_range) and repeated on the three-plane bf16 kernels -- the result is bitwise that of an engine
    created with A3R_RAFT=bf3, never a silently saturated one.  Ordinary weights do not trigger it."""
    from align3r_amd.raft import RaftEngine
    sd = synthetic_raft_state_dict(RAFT_TINY, 0)
    i1, i2 = synthetic_raft_frames(1, 128, 160, 5)
    ok = RaftEngine(RAFT_TINY, sd)
    ok.forward(dev(i1), dev(i2), iters=2)
    assert ok.fh2 and ok.range_fallbacks == 0
    big = dict(sd)
    big["cnet.layer1.0.conv1.weight"] = sd["cnet.layer1.0.conv1.weight"] * np.float32(3e5)      # context features far beyond 65504
    e2 = RaftEngine(RAFT_TINY, big)
    out = e2.forward(dev(i1), dev(i2), iters=2)
    assert e2.range_fallbacks == 1
    monkeypatch.setenv("A3R_RAFT", "bf3")
    e3 = RaftEngine(RAFT_TINY, big)
    assert not e3.fh2
    ref = e3.forward(dev(i1), dev(i2), iters=2)
    assert torch.isfinite(ref).all() and torch.equal(out, ref)


def test_raft_errors_are_loud():
    from align3r_amd.raft import RAFT2, RaftEngine
    sd = synthetic_raft_state_dict(RAFT_TINY, 0)
    eng = RaftEngine(RAFT_TINY, sd)
    with pytest.raises(RuntimeError, match="multiple of 8"):
        eng.forward(torch.zeros(1, 3, 100, 160, device="cuda"), torch.zeros(1, 3, 100, 160, device="cuda"))
    with pytest.raises(RuntimeError, match="too small"):
        eng.forward(torch.zeros(1, 3, 64, 96, device="cuda"), torch.zeros(1, 3, 64, 96, device="cuda"))
    bad = dict(sd)
    del bad["fnet.layer2.0.conv1.weight"]
    with pytest.raises(RuntimeError, match="Missing key"):
        RaftEngine(RAFT_TINY, bad)
    with pytest.raises(RuntimeError, match="HIP device"):
        RAFT2(RAFT_TINY, sd)(torch.zeros(1, 3, 128, 160), torch.zeros(1, 3, 128, 160), test_mode=True)


def test_cloud_opt_flow_computes_its_flow_with_the_hip_raft():
    """The reference's flow variant computes optical flow for every edge, both directions, inside its constructor
    (cloud_opt_flow/optimizer.py:118-154).  Same here when a flow network is given: the fields the aligner receives are exactly
    RaftEngine's outputs for (img_i * 255, img_j * 255, iters = 20) in chunks of 12, the forward-backward masks exist, and the
    flow-regularised alignment runs on them."""
    from align3r_amd.dust3r.cloud_opt_flow import global_aligner
    from align3r_amd.raft import RAFT2
    from test_gpu_api import _geom_scene
    N, H, W = 3, 128, 160
    edges, p1, p2, c, cams, depths, f = _geom_scene(N, H, W)
    a, b = synthetic_raft_frames(N, H, W, 21)
    frames = [torch.from_numpy(a[n] / 255.0 * 2 - 1) for n in range(N)]          # view['img'] is normalised to [-1, 1] (ImgNorm)
    dyn = [torch.zeros(H, W, dtype=torch.bool) for _ in range(N)]
    out = dict(view1=dict(idx=[i for i, j in edges], img=torch.stack([frames[i] for i, j in edges]), dynamic_mask=[dyn[i] for i, j in edges]),
               view2=dict(idx=[j for i, j in edges], img=torch.stack([frames[j] for i, j in edges]), dynamic_mask=[dyn[j] for i, j in edges]),
               pred1=dict(pts3d=torch.from_numpy(p1), conf=torch.from_numpy(c)),
               pred2=dict(pts3d_in_other_view=torch.from_numpy(p2), conf=torch.from_numpy(c)))
    net = RAFT2(RAFT_TINY, synthetic_raft_state_dict(RAFT_TINY, 0))
    torch.manual_seed(0)
    scene = global_aligner(out, "cuda", verbose=False, min_conf_thr=1.5, flow_loss_weight=0.01, flow_net=net, num_total_iter=20,
                           flow_loss_start_epoch=0.0)
    fij, fji = scene._flow_pair
    assert tuple(fij.shape) == (len(edges), 2, H, W) and fij.is_cuda
    imgs = np.stack(scene.imgs)                                                     # [N, H, W, 3] in [0, 1], as rgb() leaves them
    x = lambda idx: torch.from_numpy(imgs[idx]).float().permute(0, 3, 1, 2).contiguous().cuda() * 255
    ei, ej = [i for i, j in edges], [j for i, j in edges]
    want = net._engine.forward(x(ei), x(ej), iters=20)
    assert torch.equal(fij, want)
    assert torch.equal(fji, net._engine.forward(x(ej), x(ei), iters=20))
    vi = scene.flow_valid_mask_i
    assert tuple(vi.shape)[0] == len(edges) and vi.dtype in (torch.bool, torch.float32)
    loss = scene.compute_global_alignment(init="mst", niter=20, schedule="linear", lr=0.01)
    assert np.isfinite(loss)
    with pytest.raises(RuntimeError, match="no RAFT checkpoint"):
        global_aligner(out, "cuda", verbose=False, flow_loss_weight=0.01)
