"""NumPy restatement of the MaskFeat HOG target (TEST INFRASTRUCTURE ONLY).

Follows reference dataset.py:39-45 (``extract_hog_features``): per channel
``skimage.feature.hog(orientations=9, pixels_per_cell=(8,8), cells_per_block=(1,1),
block_norm='L2', feature_vector=False)`` then concat RGB and
'(ph dh)(pw dw) ch cw c -> ph pw (dh dw ch cw c)' with ph=pw=14.

scikit-image is a third-party dependency that is NOT under /root/reference
(requirements.txt:6, unpinned).  The arithmetic below restates skimage 0.18.3
(_hog.py:22-43 gradients, :179 float cast, _hoghistogram.pyx cell histogram,
_hog.py:5-11 L2 norm with eps=1e-5) and is pinned bit-exactly against the real
skimage 0.18.3 that ships under /opt/conda in the dev container
(tests/golden/make_golden.py -> tests/golden/hog_*.npz).

Exact arithmetic contract (what the HIP kernel must reproduce):
  g_row[y,x] = img[y+1,x]-img[y-1,x]   (0 on the first/last row)     float64
  g_col[y,x] = img[y,x+1]-img[y,x-1]   (0 on the first/last column)  float64
  mag = hypot(g_col, g_row); ori = rad2deg(arctan2(g_row, g_col)) % 180
  bin i takes pixels with 20*i <= ori < 20*(i+1)                     (hard assign)
  cell(r,c,i) = float32 accumulator over the 8x8 pixels in row-major order,
                each step total = (float)((double)total + mag), then /64 in float32
  block L2:   h / sqrt(sum(h^2) + 1e-10) in float64 over the 9 bins of one cell
  feature index = dh*54 + dw*27 + chan*9 + orient,  cell (2*ph+dh, 2*pw+dw)
"""
import numpy as np

ORIENT = 9
CELL = 8


def hog_bins_and_mag(chan):
    """chan: [H,W] uint8/float.  Returns (bin_id int32 [H,W] in 0..8, mag f64)."""
    img = chan.astype(np.float64)
    g_row = np.zeros_like(img)
    g_col = np.zeros_like(img)
    g_row[1:-1, :] = img[2:, :] - img[:-2, :]
    g_col[:, 1:-1] = img[:, 2:] - img[:, :-2]
    mag = np.hypot(g_col, g_row)
    ori = np.rad2deg(np.arctan2(g_row, g_col)) % 180
    bins = np.floor(ori / (180.0 / ORIENT)).astype(np.int32)
    # ori in [0,180): floor(ori/20) in 0..8; guard the (unreachable) 180.0 edge
    bins = np.minimum(bins, ORIENT - 1)
    # re-derive with the exact interval tests skimage uses, to be safe at edges
    for i in range(ORIENT):
        lo, hi = 20.0 * i, 20.0 * (i + 1)
        sel = (ori >= lo) & (ori < hi)
        bins[sel] = i
    return bins, mag


def hog_channel(chan):
    """[H,W] -> [H/8, W/8, 9] float64, L2-normalised per cell."""
    bins, mag = hog_bins_and_mag(chan)
    h, w = chan.shape
    nr, nc = h // CELL, w // CELL
    hist = np.zeros((nr, nc, ORIENT), dtype=np.float64)
    mag_c = mag[:nr * CELL, :nc * CELL].reshape(nr, CELL, nc, CELL).transpose(0, 2, 1, 3).reshape(nr, nc, CELL * CELL)
    bin_c = bins[:nr * CELL, :nc * CELL].reshape(nr, CELL, nc, CELL).transpose(0, 2, 1, 3).reshape(nr, nc, CELL * CELL)
    for i in range(ORIENT):
        tot = np.zeros((nr, nc), dtype=np.float32)
        for k in range(CELL * CELL):           # sequential float32 accumulation, row-major
            add = np.where(bin_c[:, :, k] == i, mag_c[:, :, k], 0.0)
            tot = (tot.astype(np.float64) + add).astype(np.float32)
        hist[:, :, i] = (tot / np.float32(CELL * CELL)).astype(np.float64)
    norm = np.sqrt((hist ** 2).sum(axis=-1, keepdims=True) + 1e-5 ** 2)
    return hist / norm


def extract_hog_features(image):
    """[H,W,3] uint8 -> [H/16, W/16, 108] float64 (dataset.py:39-45)."""
    per = [hog_channel(image[:, :, c]) for c in range(3)]      # 3 x [28,28,9]
    f = np.concatenate(per, axis=-1)                            # [28,28,27]
    nr, nc, k = f.shape
    f = f.reshape(nr // 2, 2, nc // 2, 2, k).transpose(0, 2, 1, 3, 4)
    return f.reshape(nr // 2, nc // 2, 4 * k)


def hog_bin_map(image):
    """[H,W,3] -> int32 [3,H,W] bin ids, for the bit-exact bin assertion."""
    return np.stack([hog_bins_and_mag(image[:, :, c])[0] for c in range(3)])
