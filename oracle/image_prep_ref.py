"""CPU restatement (numpy) of the reference's per-image preparation -- TEST INFRASTRUCTURE ONLY.

Follows /root/reference/dataset.py:37-73 (``read_images``) and :239-254 (``DiscoGANDataset._load_and_process_image``):

    image = np.array(Image.open(fn).convert('RGB'))            # uint8 [H, W, 3]                      :44,49 / :242-243
    domain 'A':  image = image[:, :256, :]                      # left half of an edges2* pair          :54 / :248
                 image = 255. - image                           # float64 from here on                  :55 / :249
                 image = cv2.dilate(image, ones((3,3)), 1)      # 3x3 max, border pixels do not count   :56 / :250
                 image = 255. - image                           # => a 3x3 EROSION of the uint8 values  :57 / :251
    domain 'B':  image = image[:, 256:, :]                      # right half, stays uint8               :59 / :253
    image = cv2.resize(image, (S, S))                           # INTER_LINEAR                          :62 / :256
    image = image.astype(np.float32) / 255.                     #                                       :65 / :257
    image = image.transpose(2, 0, 1)                            # CHW                                   :66 / :258

The arithmetic of cv2.resize lives in OpenCV (third party: the reference pins ``opencv-python`` only by name in
requirements.txt; cv2 is NOT installed in this image), so this file restates OpenCV's published INTER_LINEAR algorithm
(modules/imgproc/src/resize.cpp): sample position ``(d + 0.5) * scale - 0.5`` computed in double and cast to float,
floor, clamp to the first / last source pixel with weight 0; then
  * float images (the float64 domain-'A' image): coefficients in fp32, horizontal pass, vertical pass;
  * uint8 images: 11-bit fixed point -- coefficients round(2048 a) (int16), rows R = S0 a0 + S1 a1 (int32), output
    ``(((b0 (R0 >> 4)) >> 16) + ((b1 (R1 >> 4)) >> 16) + 2) >> 2`` (VResizeLinear<uchar,int,short>), a uint8.
PARITY UNPINNED for the resize step against cv2 itself (cv2 absent here; nothing in the reference's tests holds a resized
golden image).  What IS pinned: the sampling convention against torch.nn.functional.interpolate(bilinear,
align_corners=False) on CPU (tests/test_ingest_cpu.py), the erosion against scipy.ndimage, and crop / normalise / CHW exactly.
Only tests/ (and the smoke check) import this module; the product path is csrc/ingest.hip.
"""
import numpy as np


def crop_domain(image_u8, domain):
    """dataset.py:52-59: 'A' -> left 256 columns, 'B' -> columns 256.., None -> the whole image."""
    if domain == "A":
        return image_u8[:, :256, :]
    if domain == "B":
        return image_u8[:, 256:, :]
    return image_u8


def erode3x3(image_u8):
    """255 - cv2.dilate(255 - x, ones((3,3))): minimum over the in-bounds 3x3 neighbourhood (cv2's default border value for
    dilate makes outside pixels lose every comparison).  Integer-valued float64, as the reference has it."""
    h, w, _ = image_u8.shape
    x = image_u8.astype(np.float64)
    out = np.full_like(x, 255.0)
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            ys, ye = max(0, -dy), min(h, h - dy)
            xs, xe = max(0, -dx), min(w, w - dx)
            out[ys:ye, xs:xe] = np.minimum(out[ys:ye, xs:xe], x[ys + dy:ye + dy, xs + dx:xe + dx])
    return out


def _axis(n_src, n_dst):
    """(first tap index, weight of the second tap as float32) per destination index -- cv2 resize.cpp's table."""
    scale = n_src / n_dst
    d = np.arange(n_dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = f - s.astype(np.float32)
    lo = s < 0
    f[lo], s[lo] = 0.0, 0
    hi = s >= n_src - 1
    f[hi], s[hi] = 0.0, n_src - 1
    return s, f.astype(np.float32)


def resize_linear_float(image, size):
    """INTER_LINEAR of a float image [H, W, C] to [size, size, C]; fp32 coefficients, horizontal then vertical pass."""
    h, w, _ = image.shape
    sx, fx = _axis(w, size)
    sy, fy = _axis(h, size)
    x1 = np.minimum(sx + 1, w - 1)
    y1 = np.minimum(sy + 1, h - 1)
    img = image.astype(np.float64)
    a1 = fx.astype(np.float64)[None, :, None]
    a0 = (np.float32(1.0) - fx).astype(np.float64)[None, :, None]
    rows = img[:, sx, :] * a0 + img[:, x1, :] * a1                    # [H, size, C]
    b1 = fy.astype(np.float64)[:, None, None]
    b0 = (np.float32(1.0) - fy).astype(np.float64)[:, None, None]
    return rows[sy] * b0 + rows[y1] * b1


def resize_linear_u8(image_u8, size):
    """INTER_LINEAR of a uint8 image: OpenCV's 11-bit fixed-point path (see the module docstring); returns uint8."""
    h, w, _ = image_u8.shape
    sx, fx = _axis(w, size)
    sy, fy = _axis(h, size)
    x1 = np.minimum(sx + 1, w - 1)
    y1 = np.minimum(sy + 1, h - 1)
    a1 = np.rint(fx * np.float32(2048.0)).astype(np.int64)[None, :, None]
    a0 = np.rint((np.float32(1.0) - fx) * np.float32(2048.0)).astype(np.int64)[None, :, None]
    b1 = np.rint(fy * np.float32(2048.0)).astype(np.int64)[:, None, None]
    b0 = np.rint((np.float32(1.0) - fy) * np.float32(2048.0)).astype(np.int64)[:, None, None]
    img = image_u8.astype(np.int64)
    rows = img[:, sx, :] * a0 + img[:, x1, :] * a1                    # int32 range
    v = (((b0 * (rows[sy] >> 4)) >> 16) + ((b1 * (rows[y1] >> 4)) >> 16) + 2) >> 2
    return np.clip(v, 0, 255).astype(np.uint8)


def prepare_image(image_u8, domain, image_size):
    """One image through dataset.py:52-66 -> float32 [3, S, S]."""
    img = crop_domain(np.asarray(image_u8), domain)
    if domain == "A":
        out = resize_linear_float(erode3x3(img), image_size)          # float64 image: unrounded result
    else:
        out = resize_linear_u8(img, image_size)                       # uint8 image: rounded to uint8
    out = out.astype(np.float32) / np.float32(255.0)
    return np.ascontiguousarray(out.transpose(2, 0, 1))


def read_images(images_u8, domain=None, image_size=64):
    """dataset.py:37-73 on already-decoded arrays: stack of prepared images [n, 3, S, S] float32."""
    return np.stack([prepare_image(im, domain, image_size) for im in images_u8])
