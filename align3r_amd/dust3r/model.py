"""AsymmetricCroCo3DStereo behind the reference's Python API (dust3r/model.py:27-257), computed by liba3r.

Same constructor keywords, checkpoint format ({'args': Namespace(model=<str>), 'model': state_dict}),
state-dict key names, ``load_state_dict`` duplication rule (dec_blocks -> dec_blocks2, model.py:114-121)
and forward contract: ``forward(view1, view2) -> (res1, res2)`` with res1 = {pts3d, conf, pred_mask},
res2 = {pts3d_in_other_view, conf, pred_mask}.  There is no CPU compute path: forward on a model that
has not been moved to a HIP device raises.
Differences, on purpose: the checkpoint's model string is PARSED (ast), never eval'd (model.py:39 evals it);
checkpoints are read with torch.load(weights_only=True); the HuggingFace-hub branch of from_pretrained
(model.py:105) is not available offline and raises.
"""
from __future__ import annotations

import argparse
import ast
import os
from collections import namedtuple
from typing import Dict

import numpy as np
import torch

from ..weights import ModelConfig, param_spec, reference_aliases, synthetic_state_dict

inf = float('inf')
_IncompatibleKeys = namedtuple('_IncompatibleKeys', ['missing_keys', 'unexpected_keys'])


def _parse_model_string(s: str) -> dict:
    """'AsymmetricCroCo3DStereo(k=v, ...)' -> kwargs, literals only (inf allowed)."""
    tree = ast.parse(s.strip(), mode='eval').body
    if not (isinstance(tree, ast.Call) and getattr(tree.func, 'id', None) == 'AsymmetricCroCo3DStereo' and not tree.args):
        raise ValueError(f'unsupported model string: {s!r}')

    def lit(node):
        if isinstance(node, ast.Name) and node.id == 'inf':
            return inf
        if isinstance(node, ast.UnaryOp) and isinstance(node.op, ast.USub):
            return -lit(node.operand)
        if isinstance(node, ast.Tuple):
            return tuple(lit(e) for e in node.elts)
        if isinstance(node, ast.List):
            return [lit(e) for e in node.elts]
        return ast.literal_eval(node)
    return {kw.arg: lit(kw.value) for kw in tree.keywords}


def load_model(model_path, device, verbose=True):
    """dust3r/model.py:27-43 (same patching of the model string: PatchEmbedDust3R, landscape_only=False)."""
    if verbose:
        print('... loading model from', model_path)
    with torch.serialization.safe_globals([argparse.Namespace]):
        ckpt = torch.load(model_path, map_location='cpu', weights_only=True)
    args = ckpt['args'].model.replace("ManyAR_PatchEmbed", "PatchEmbedDust3R")
    kwargs = _parse_model_string(args)
    kwargs['landscape_only'] = False
    if verbose:
        print(f"instantiating : AsymmetricCroCo3DStereo({kwargs})")
    net = AsymmetricCroCo3DStereo(**kwargs)
    s = net.load_state_dict(ckpt['model'], strict=False)
    if verbose:
        print(s)
    return net.to(device)


class AsymmetricCroCo3DStereo:
    """Two siamese encoders + two cross-attending decoders + depth-prior side branch + two DPT heads."""

    def __init__(self, output_mode='pts3d', head_type='linear', depth_mode=('exp', -inf, inf),
                 conf_mode=('exp', 1, inf), freeze='none', landscape_only=True, patch_embed_cls='PatchEmbedDust3R',
                 img_size=224, patch_size=16, mask_ratio=0.9, enc_embed_dim=768, enc_depth=12, enc_num_heads=12,
                 dec_embed_dim=512, dec_depth=8, dec_num_heads=16, mlp_ratio=4, norm_im2_in_dec=True, pos_embed='cosine'):
        if head_type != 'dpt':
            raise NotImplementedError("only head_type='dpt' (the reference fork's linear head is broken: linear_head.py:35)")
        if output_mode != 'pts3d':
            raise NotImplementedError(f'{output_mode=}')
        if tuple(depth_mode) != ('exp', -inf, inf) or tuple(conf_mode) != ('exp', 1, inf):
            raise NotImplementedError("only depth_mode=('exp',-inf,inf), conf_mode=('exp',1,inf) (train.sh:7)")
        if not (isinstance(pos_embed, str) and pos_embed.startswith('RoPE')):
            raise NotImplementedError("only RoPE positional embedding (pos_embed='RoPE100')")
        if patch_embed_cls not in ('PatchEmbedDust3R', 'ManyAR_PatchEmbed'):
            raise AssertionError(patch_embed_cls)
        if not norm_im2_in_dec or int(mlp_ratio) != mlp_ratio:
            raise NotImplementedError('norm_im2_in_dec=False / fractional mlp_ratio')
        img_size = (img_size, img_size) if isinstance(img_size, int) else tuple(img_size)
        assert img_size[0] % patch_size == 0 and img_size[1] % patch_size == 0, \
            f'{img_size=} must be multiple of {patch_size=}'
        self.cfg = ModelConfig(enc_embed_dim=enc_embed_dim, enc_depth=enc_depth, enc_num_heads=enc_num_heads,
                               dec_embed_dim=dec_embed_dim, dec_depth=dec_depth, dec_num_heads=dec_num_heads,
                               mlp_ratio=int(mlp_ratio), patch_size=patch_size, rope_base=float(pos_embed[len('RoPE'):]))
        self.output_mode, self.head_type = output_mode, head_type
        self.depth_mode, self.conf_mode = depth_mode, conf_mode
        self.landscape_only = landscape_only
        self.patch_embed_cls = patch_embed_cls
        self.enc_depth, self.enc_embed_dim = enc_depth, enc_embed_dim
        self.dec_depth, self.dec_embed_dim = dec_depth, dec_embed_dim
        self.freeze = freeze
        self.training = False
        self.device = torch.device('cpu')
        self._params: Dict[str, torch.Tensor] | None = None
        self._engine = None

    # ------------------------------------------------------------------ weights
    def _ensure_params(self):
        if self._params is None:   # the reference draws torch-RNG init weights here; we use the deterministic generator
            self._params = {k: torch.from_numpy(v) for k, v in synthetic_state_dict(self.cfg, 0).items()}

    def state_dict(self):
        self._ensure_params()
        sd = dict(self._params)
        for alias, canon in reference_aliases(self.cfg).items():
            sd[alias] = sd[canon]
        return sd

    def load_state_dict(self, ckpt, strict=True, **kw):
        new_ckpt = dict(ckpt)
        if not any(k.startswith('dec_blocks2') for k in ckpt):        # model.py:114-121
            for key, value in ckpt.items():
                if key.startswith('dec_blocks') and not key.startswith('dec_blocks_pc'):
                    new_ckpt[key.replace('dec_blocks', 'dec_blocks2')] = value
        spec = {n: s for n, s, _ in param_spec(self.cfg)}
        aliases = reference_aliases(self.cfg)
        missing = [k for k in spec if k not in new_ckpt]
        unexpected = [k for k in new_ckpt if k not in spec and k not in aliases]
        errors = []
        loaded = {}
        for k, shp in spec.items():
            if k in new_ckpt:
                t = torch.as_tensor(new_ckpt[k]).detach().to('cpu', torch.float32)
                if tuple(t.shape) != tuple(shp):
                    errors.append(f'size mismatch for {k}: copying a param with shape {tuple(t.shape)} from checkpoint, '
                                  f'the shape in current model is {tuple(shp)}.')
                else:
                    loaded[k] = t.contiguous()
        if strict and (missing or unexpected):
            errors.append(f'Missing key(s): {missing}. Unexpected key(s): {unexpected}.')
        if errors:
            raise RuntimeError('Error(s) in loading state_dict for AsymmetricCroCo3DStereo:\n\t' + '\n\t'.join(errors))
        if missing:
            self._ensure_params()
        self._params = {**(self._params or {}), **loaded}
        self._engine = None
        if self.device.type == 'cuda':
            self.to(self.device)
        return _IncompatibleKeys(missing, unexpected)

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, **kw):
        if os.path.isfile(pretrained_model_name_or_path):
            return load_model(pretrained_model_name_or_path, device='cpu')
        raise Exception(f'tried to load {pretrained_model_name_or_path} from huggingface, but failed '
                        '(no network access in this build: pass a local checkpoint file)')

    # ------------------------------------------------------------------ nn.Module-ish surface
    def to(self, device):
        device = torch.device(device)
        if device.type == 'cuda' and device.index is None:
            device = torch.device('cuda', torch.cuda.current_device())
        self.device = device
        if device.type == 'cuda':
            from ..engine import PairEngine
            self._ensure_params()
            self._engine = PairEngine(self.cfg, self._params, device)
        else:
            self._engine = None
        return self

    def cuda(self, device=None):
        return self.to('cuda' if device is None else device)

    def eval(self):
        self.training = False
        return self

    def train(self, mode=True):
        if mode:
            raise NotImplementedError('training is out of scope (SURVEY.md section 2); inference engine only')
        return self

    def parameters(self):
        self._ensure_params()
        return iter(self._params.values())

    def __call__(self, view1, view2, out=None):
        return self.forward(view1, view2, out=out)

    # ------------------------------------------------------------------ forward (model.py:241-257)
    def forward(self, view1, view2, out=None):
        """out (extension): dict(pts3d_1 [B,H,W,3], conf_1 [B,H,W], pts3d_2, conf_2) of contiguous float32 device tensors the engine
        writes its results into -- e.g. this rank's rows of the gathered aligner buffers (parallel.sharded_inference)."""
        img1, img2 = view1['img'], view2['img']
        if img1.shape[-2:] != img2.shape[-2:]:
            # The encoder of the reference still takes the two sizes apart (model.py:171-173, inherited from DUSt3R), but its
            # forward then concatenates the two views' point maps along the batch axis (model.py:248, the Align3R addition), which
            # torch.cat refuses for different H x W: the reference cannot run such a pair either.  Same exception type, its message.
            raise RuntimeError(f'Sizes of tensors must match except in dimension 0. Expected size {img1.shape[-2]}x{img1.shape[-1]} but got '
                               f'size {img2.shape[-2]}x{img2.shape[-1]} (the two views of a pair must have one image size: the '
                               'reference concatenates their point maps, dust3r/model.py:248)')
        if self._engine is None:
            raise RuntimeError('AsymmetricCroCo3DStereo.forward: the model is not on a HIP device -- call .to("cuda"); '
                               'this build has no CPU compute path')
        dev = self.device
        B, _, H, W = img1.shape
        for v in (view1, view2):       # utils/misc.py:61: all true_shape identical when landscape_only=False
            ts = v.get('true_shape')
            if ts is not None:
                ts = torch.as_tensor(ts).reshape(-1, 2)
                assert bool((ts == ts[0:1]).all()), 'true_shape must be all identical'
                assert tuple(int(x) for x in ts[0]) == (H, W), 'true_shape must match the image size'
        if self.landscape_only:
            assert W >= H, f'img should be in landscape mode, but got {W=} {H=}'
        f = lambda t: t.to(dev, torch.float32).contiguous()
        out = self._engine.forward(f(img1), f(img2), f(view1['pred_depth']), f(view2['pred_depth']), out=out)
        res1 = dict(pts3d=out['pts3d_1'], conf=out['conf_1'], pred_mask=0)
        res2 = dict(pts3d_in_other_view=out['pts3d_2'], conf=out['conf_2'], pred_mask=0)
        return res1, res2


    # ------------------------------------------------------------------ encoder feature caching (extension)
    def encode_frames(self, imgs):
        """_encode_image (model.py:151-163) for a batch of frames [B,3,H,W]: enc_norm'd tokens [B,N,enc_embed_dim].
        The encoder output depends on the frame only, so a clip needs it once per frame, not once per pair."""
        if self._engine is None:
            raise RuntimeError('the model is not on a HIP device -- call .to("cuda")')
        return self._engine.encode(imgs.to(self.device, torch.float32).contiguous())

    def forward_cached(self, view1, view2, feat1, feat2):
        """forward() with the two views' encoder features given (same outputs, bit-identical)."""
        if self._engine is None:
            raise RuntimeError('the model is not on a HIP device -- call .to("cuda")')
        H, W = view1['img'].shape[-2:]
        f = lambda t: t.to(self.device, torch.float32).contiguous()
        out = self._engine.decode(feat1, feat2, f(view1['pred_depth']), f(view2['pred_depth']), H, W)
        return (dict(pts3d=out['pts3d_1'], conf=out['conf_1'], pred_mask=0),
                dict(pts3d_in_other_view=out['pts3d_2'], conf=out['conf_2'], pred_mask=0))


def save_checkpoint(path, model: AsymmetricCroCo3DStereo, img_size=(512, 512), epoch=0):
    """Write a reference-format checkpoint (croco/utils/misc.py:292-305) for `model`."""
    from ..weights import model_string
    torch.save({'args': argparse.Namespace(model=model_string(model.cfg, img_size)), 'model': model.state_dict(),
                'epoch': epoch}, path)
