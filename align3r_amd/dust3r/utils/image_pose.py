"""Input preprocessing of the pair forward: what turns image files + mono-depth .npz into view dicts.

Mirror of the reference's dust3r/utils/image_pose.py (same names, arguments and results) without its cv2 / torchvision /
imageio dependencies:
  ImgNorm, ToTensor                      image_pose.py:26-27    (torchvision ToTensor + Normalize(0.5, 0.5) restated)
  _resize_pil_image, crop_img            image_pose.py:112-118, 172-204   (PIL, identical calls)
  resize_numpy_image, crop_center        image_pose.py:120-170  (cv2.resize restated, see below)
  pixel_to_pointcloud, normalize_pointcloud   image_pose.py:206-244
  load_images                            image_pose.py:246-407  (image files; video needs cv2.VideoCapture: not available)
  depth_read, cam_read, flow_read        image_pose.py:30-72    (Sintel binary formats)

Parity: everything that is PIL / numpy arithmetic is pinned bit for bit against the reference (tests/golden/prep.npz).
`resize_numpy_image` resizes the un-projected mono point map with cv2.resize (INTER_LANCZOS4 when shrinking, INTER_CUBIC
otherwise); OpenCV is not installed in this image, so its two filters are restated here from their published
definitions (pixel-centre alignment, 8-tap Lanczos a=4 / 4-tap cubic A=-0.75, weights normalised, BORDER_REPLICATE, no
anti-aliasing) and that function is PARITY UNPINNED.
"""
from __future__ import annotations

import os

import numpy as np
import PIL.Image
import torch
from PIL.ImageOps import exif_transpose

TAG_FLOAT = 202021.25


def ToTensor(pic):
    """torchvision.transforms.ToTensor for PIL 'RGB' / 'L' images: uint8 HWC -> float32 CHW in [0, 1]."""
    arr = np.array(pic, copy=True)
    if arr.ndim == 2:
        arr = arr[:, :, None]
    t = torch.from_numpy(arr).permute(2, 0, 1).contiguous()
    return t.to(torch.float32).div(255) if t.dtype == torch.uint8 else t


def ImgNorm(pic):
    """tvf.Compose([ToTensor(), Normalize((0.5,)*3, (0.5,)*3)])  (image_pose.py:26)."""
    t = ToTensor(pic)
    mean = torch.as_tensor((0.5, 0.5, 0.5), dtype=t.dtype).view(-1, 1, 1)
    std = torch.as_tensor((0.5, 0.5, 0.5), dtype=t.dtype).view(-1, 1, 1)
    return t.sub_(mean).div_(std)


def _read_tagged(f, what):
    check = np.fromfile(f, dtype=np.float32, count=1)[0]
    assert check == TAG_FLOAT, f' {what}:: Wrong tag in flow file (should be: {TAG_FLOAT}, is: {check}). Big-endian machine? '


def depth_read(filename):
    """Sintel .dpt (image_pose.py:30-41)."""
    with open(filename, 'rb') as f:
        _read_tagged(f, 'depth_read')
        width = np.fromfile(f, dtype=np.int32, count=1)[0]
        height = np.fromfile(f, dtype=np.int32, count=1)[0]
        size = width * height
        assert width > 0 and height > 0 and size > 1 and size < 100000000, f' depth_read:: Wrong input size (width = {width}, height = {height}).'
        return np.fromfile(f, dtype=np.float32, count=-1).reshape((height, width))


def cam_read(filename):
    """Sintel .cam -> (M intrinsics 3x3, N extrinsics 3x4) (image_pose.py:43-57)."""
    with open(filename, 'rb') as f:
        _read_tagged(f, 'cam_read')
        M = np.fromfile(f, dtype='float64', count=9).reshape((3, 3))
        N = np.fromfile(f, dtype='float64', count=12).reshape((3, 4))
        return M, N


def flow_read(filename):
    """Middlebury .flo -> (u, v) (image_pose.py:59-74)."""
    with open(filename, 'rb') as f:
        _read_tagged(f, 'flow_read')
        width = np.fromfile(f, dtype=np.int32, count=1)[0]
        height = np.fromfile(f, dtype=np.int32, count=1)[0]
        size = width * height
        assert width > 0 and height > 0 and size > 1 and size < 100000000, f' flow_read:: Wrong input size (width = {width}, height = {height}).'
        tmp = np.fromfile(f, dtype=np.float32, count=-1).reshape((height, width * 2))
        return tmp[:, np.arange(width) * 2], tmp[:, np.arange(width) * 2 + 1]


def rgb(ftensor, true_shape=None):
    """Display form of a network image (image_pose.py:93-110): channels last, optionally cut to true_shape = (H, W),
    uint8 scaled by 1/255 and ImgNorm'ed floats mapped back from [-1, 1], everything clipped to [0, 1].  Lists map elementwise."""
    if isinstance(ftensor, list):
        return [rgb(item, true_shape=true_shape) for item in ftensor]
    arr = ftensor.detach().cpu().numpy() if isinstance(ftensor, torch.Tensor) else ftensor
    channels_first = {3: 0, 4: 1}.get(arr.ndim)                 # CHW or BCHW with 3 channels -> HWC / BHWC
    if channels_first is not None and arr.shape[channels_first] == 3:
        arr = np.moveaxis(arr, channels_first, -1)
    if true_shape is not None:
        arr = arr[:true_shape[0], :true_shape[1]]
    scaled = arr.astype(np.float32) / 255 if arr.dtype == np.uint8 else arr * 0.5 + 0.5
    return np.clip(scaled, 0, 1)


def _resize_pil_image(img, long_edge_size, nearest=False):
    """Long edge -> long_edge_size (image_pose.py:112-118): LANCZOS (or NEAREST) when shrinking, BICUBIC when enlarging."""
    longest = max(img.size)
    scale = long_edge_size / longest
    if longest <= long_edge_size:
        resample = PIL.Image.BICUBIC
    else:
        resample = PIL.Image.NEAREST if nearest else PIL.Image.LANCZOS
    return img.resize((int(round(img.size[0] * scale)), int(round(img.size[1] * scale))), resample)


def _lanczos4_weights(frac):
    """8 taps at offsets -3..4 around floor(x): sinc(t) sinc(t/4), normalised to sum 1 (OpenCV interpolateLanczos4)."""
    t = frac[:, None] - np.arange(-3, 5, dtype=np.float64)[None, :]
    w = np.sinc(t) * np.sinc(t / 4.0)
    return w / w.sum(1, keepdims=True)


def _cubic_weights(frac, A=-0.75):
    """4 taps at offsets -1..2 (OpenCV interpolateCubic, A = -0.75)."""
    x = frac
    w = np.empty((len(x), 4), np.float64)
    w[:, 0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A
    w[:, 1] = ((A + 2) * x - (A + 3)) * x * x + 1
    w[:, 2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1
    w[:, 3] = 1.0 - w[:, 0] - w[:, 1] - w[:, 2]
    return w


def _resize_axis(img, new_len, axis, lanczos):
    n = img.shape[axis]
    scale = n / float(new_len)
    fx = (np.arange(new_len, dtype=np.float64) + 0.5) * scale - 0.5        # pixel-centre alignment
    sx = np.floor(fx)
    frac = fx - sx
    if lanczos:
        w, offs = _lanczos4_weights(frac), np.arange(-3, 5)
    else:
        w, offs = _cubic_weights(frac), np.arange(-1, 3)
    idx = np.clip(sx[:, None].astype(np.int64) + offs[None, :], 0, n - 1)   # BORDER_REPLICATE
    src = np.moveaxis(img, axis, 0).astype(np.float64)
    out = np.zeros((new_len,) + src.shape[1:], np.float64)
    for k in range(w.shape[1]):
        out += src[idx[:, k]] * w[:, k].reshape((-1,) + (1,) * (src.ndim - 1))
    return np.moveaxis(out, 0, axis)


def cv2_resize(img, new_size, lanczos):
    """cv2.resize(img, (w, h), interpolation=INTER_LANCZOS4 | INTER_CUBIC) on a float array [H, W(, C)] (parity unpinned)."""
    w, h = new_size
    out = _resize_axis(img, w, 1, lanczos)
    out = _resize_axis(out, h, 0, lanczos)
    return out.astype(img.dtype)


def resize_numpy_image(img, long_edge_size):
    """image_pose.py:120-147: long edge -> long_edge_size, LANCZOS4 when shrinking, CUBIC otherwise."""
    h, w = img.shape[:2]
    S = max(h, w)
    new_size = (int(round(w * long_edge_size / S)), int(round(h * long_edge_size / S)))
    return cv2_resize(img, new_size, lanczos=S > long_edge_size)


def crop_center(img, crop_width, crop_height):
    """Window of crop_height rows x crop_width columns around the array centre, clipped to the array (image_pose.py:149-170)."""
    rows, cols = img.shape[:2]
    r_mid, c_mid = rows // 2, cols // 2
    r_half, c_half = crop_height // 2, crop_width // 2
    return img[max(r_mid - r_half, 0):min(r_mid + r_half, rows), max(c_mid - c_half, 0):min(c_mid + c_half, cols)]


def crop_img(img, size, pred_depth=None, square_ok=False, nearest=False, crop=True):
    """image_pose.py:172-204.  size == 224: short side -> 224, centre square.  Otherwise: long side -> size, then the centre
    window whose sides are multiples of 16 (4:3 for square inputs unless square_ok), cut out (crop) or squeezed into (not crop).
    The point map `pred_depth` [H, W, 3] follows the image."""
    W1, H1 = img.size
    long_edge = round(size * max(W1 / H1, H1 / W1)) if size == 224 else size
    img = _resize_pil_image(img, long_edge, nearest=nearest)
    if pred_depth is not None:
        pred_depth = resize_numpy_image(pred_depth, long_edge)
    W, H = img.size
    cx, cy = W // 2, H // 2
    if size == 224:
        halfw = halfh = min(cx, cy)
    else:
        halfw, halfh = ((2 * cx) // 16) * 8, ((2 * cy) // 16) * 8
        if W == H and not square_ok:
            halfh = 3 * halfw / 4
        if not crop:
            target = (2 * halfw, 2 * halfh)
            img = img.resize(target, PIL.Image.LANCZOS)
            if pred_depth is not None:
                pred_depth = cv2_resize(pred_depth, target, lanczos=False)
            return img, pred_depth
    img = img.crop((cx - halfw, cy - halfh, cx + halfw, cy + halfh))
    if pred_depth is not None:
        pred_depth = crop_center(pred_depth, 2 * halfw, 2 * halfh)
    return img, pred_depth


def normalize_pointcloud(point_cloud):
    """image_pose.py:239-244: per-channel min-max to [0, 1]."""
    min_vals = np.min(point_cloud, axis=(0, 1))
    max_vals = np.max(point_cloud, axis=(0, 1))
    return (point_cloud - min_vals) / (max_vals - min_vals)


def pixel_to_pointcloud(depth_map, focal_length_px):
    """image_pose.py:206-237: un-project a depth map with a pinhole centred at (W/2, H/2), then min-max normalise -> [H, W, 3] float32.
    (float64 grid arithmetic exactly as the reference's numpy expression order: the result is pinned bit for bit, prep.npz.)"""
    rows, cols = depth_map.shape
    px, py = np.meshgrid(np.arange(cols), np.arange(rows))
    x_cam = (px - cols / 2) * depth_map / focal_length_px
    y_cam = (py - rows / 2) * depth_map / focal_length_px
    return normalize_pointcloud(np.stack((x_cam, y_cam, depth_map), axis=-1).astype(np.float32))


_PRIOR_PATH_RULES = {
    'sintel': lambda p, n: p.replace('clean', 'depth_prediction_' + n).replace('.png', '.npz'),
    'tum': lambda p, n: p.replace('rgb_50', 'rgb_50_depth_prediction_' + n).replace('.png', '.npz'),
    'tartanair': lambda p, n: p.replace('rgb_50', 'rgb_50_depth_prediction_' + n).replace('.png', '.npz'),
    'bonn': lambda p, n: p.replace('rgb_110', 'rgb_110_depth_prediction_' + n).replace('.png', '.npz'),
    'davis': lambda p, n: p.replace('JPEGImages', 'depth_prediction_' + n).replace('.jpg', '.npz').replace('480p', '1080p'),
    'scannet': lambda p, n: p.replace('color_30', 'color_90_depth_prediction_' + n).replace('.jpg', '.npz').replace('.png', '.npz'),
    'kitti': lambda p, n: p.replace('image_gathered', 'depth_prediction_' + n).replace('.jpg', '.npz').replace('.png', '.npz'),
}


def depth_prior_path(full_path, traj_format, depth_prior_name):
    """Where load_images looks for the mono-depth prior of an image (image_pose.py:292-305)."""
    rule = _PRIOR_PATH_RULES.get(traj_format)
    if rule is not None:
        return rule(full_path, depth_prior_name)
    return full_path.replace('.png', '_pred_depth_' + depth_prior_name + '.npz').replace('.jpg', '_pred_depth_' + depth_prior_name + '.npz')


def _list_inputs(folder_or_list, verbose):
    if isinstance(folder_or_list, str):
        if verbose:
            print(f'>> Loading images from {folder_or_list}')
        if os.path.isdir(folder_or_list):
            return folder_or_list, sorted(os.listdir(folder_or_list))
        return '', [folder_or_list]
    if isinstance(folder_or_list, list):
        if verbose:
            print(f'>> Loading a list of {len(folder_or_list)} items')
        return '', folder_or_list
    raise ValueError(f'Bad input {folder_or_list=} ({type(folder_or_list)})')


def _dynamic_mask_for(full_path, name, size, square_ok, dynamic_mask_root, like):
    """Ground-truth motion mask next to the image if there is one (Sintel layout by default), else all-static."""
    if dynamic_mask_root is not None:
        mask_path = os.path.join(dynamic_mask_root, os.path.basename(name))
    else:
        mask_path = full_path.replace('final', 'dynamic_label_perfect').replace('clean', 'dynamic_label_perfect') \
            .replace('MPI-Sintel-training_images', 'MPI-Sintel-depth-training')
    if not os.path.exists(mask_path):
        return torch.zeros_like(like)
    m, _ = crop_img(PIL.Image.open(mask_path).convert('L'), size, square_ok=square_ok)
    return ToTensor(m)[None].sum(1) > 0.99          # "1" means dynamic


def load_images(folder_or_list, size, square_ok=False, verbose=True, dynamic_mask_root=None, crop=True, fps=0, traj_format="sintel",
                start=0, interval=30, depth_prior_name='depthpro'):
    """Image files (+ their mono-depth .npz priors) -> the view dicts of the pair forward (image_pose.py:246-407).

    Returns (imgs, imgs_raw): imgs[k] = dict(img [1,3,H,W] in [-1,1], pred_depth [1,H,W,3] in [0,1], true_shape int32 [1,2],
    idx, instance, mask, dynamic_mask).  Files are taken in name order, `interval` of them from `start`."""
    root, names = _list_inputs(folder_or_list, verbose)
    names = sorted(names, key=lambda x: x.split('/')[-1])[start: start + interval]
    imgs, imgs_raw = [], []
    for name in names:
        full_path = os.path.join(root, name)
        low = name.lower()
        if low.endswith(('.mp4', '.avi', '.mov')):
            raise NotImplementedError(f'{full_path}: video decoding needs cv2.VideoCapture (image_pose.py:340-401), which is not '
                                      'available; extract the frames to image files first')
        if not low.endswith(('.jpg', '.jpeg', '.png')):
            continue
        raw = exif_transpose(PIL.Image.open(full_path)).convert('RGB')
        imgs_raw.append(raw)
        prior = np.load(depth_prior_path(full_path, traj_format, depth_prior_name))       # allow_pickle stays False
        focal_px = prior['focallength_px'] if depth_prior_name == 'depthpro' else 200
        depth = prior['depth']
        if depth.ndim == 3:
            depth = np.squeeze(depth)
        img, pointmap = crop_img(raw, size, pixel_to_pointcloud(depth, focal_px), square_ok=square_ok, crop=crop)
        if verbose:
            print(f' - Adding {name} with resolution {raw.size[0]}x{raw.size[1]} --> {img.size[0]}x{img.size[1]}')
        view = dict(img=ImgNorm(img)[None], pred_depth=pointmap[None, ...], true_shape=np.int32([img.size[::-1]]), idx=len(imgs),
                    instance=full_path, mask=~(ToTensor(img)[None].sum(1) <= 0.01))
        view['dynamic_mask'] = _dynamic_mask_for(full_path, name, size, square_ok, dynamic_mask_root, view['mask'])
        imgs.append(view)
    assert imgs, 'No images found at ' + root
    if verbose:
        print(f' (Found {len(imgs)} images)')
    return imgs, imgs_raw
