"""HOG oracles (NumPy and C restatements) against real scikit-image 0.18.3 golden vectors."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from helpers import gold, ROOT
from oracle import hog_oracle as H


def _frames():
    fr = [np.random.RandomState(s).randint(0, 256, (224, 224, 3)).astype(np.uint8) for s in (1234, 7)]
    yy, xx = np.mgrid[0:224, 0:224]
    fr.append(np.stack([(yy * 255 // 223), (xx * 255 // 223), ((xx * 3 + yy * 5) // 8 % 256)], -1).astype(np.uint8))
    return fr


def _clib():
    so = os.path.join(ROOT, 'oracle', '_build', 'libhogref.so')
    src = os.path.join(ROOT, 'oracle', 'hog_ref.c')
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(so), exist_ok=True)
        subprocess.run(['gcc', '-O2', '-ffp-contract=off', '-shared', '-fPIC', '-o', so, src, '-lm'], check=True)
    lib = ctypes.CDLL(so)
    lib.vtx_ref_hog_bin.restype = ctypes.c_int
    return lib


def test_numpy_oracle_bit_exact_vs_skimage():
    feats = gold('hog_skimage.npz')['feats']
    for f, ref in zip(_frames(), feats):
        got = H.extract_hog_features(f)
        assert got.dtype == np.float64 and got.shape == (14, 14, 108)
        assert np.array_equal(got, ref)
    # known-answer from SURVEY.md section 8(c)
    assert abs(feats[0].sum() - 6520.430081395043) < 1e-9


def test_c_oracle_bit_exact_vs_skimage():
    lib = _clib()
    feats = gold('hog_skimage.npz')['feats']
    for f, ref in zip(_frames(), feats):
        out = np.zeros((14, 14, 108))
        bins = np.zeros((3, 224, 224), np.int32)
        lib.vtx_ref_hog_frame(f.ctypes.data_as(ctypes.c_void_p), 224, 224, out.ctypes.data_as(ctypes.c_void_p),
                              bins.ctypes.data_as(ctypes.c_void_p))
        assert np.array_equal(out, ref)
        assert np.array_equal(bins, H.hog_bin_map(f))


def test_sign_test_bins_equal_atan2_bins_exhaustively():
    """Every integer gradient pair a uint8 image can produce: the exact sign-test bin (what the
    HIP kernel evaluates) equals skimage's atan2-based interval test."""
    lib = _clib()
    g = np.arange(-255, 256)
    GR, GC = np.meshgrid(g, g, indexing='ij')
    ori = np.rad2deg(np.arctan2(GR.astype(np.float64), GC.astype(np.float64))) % 180
    want = np.zeros_like(GR)
    for i in range(9):
        want[(ori >= 20 * i) & (ori < 20 * (i + 1))] = i
    got = np.array([[lib.vtx_ref_hog_bin(int(a), int(b)) for b in g] for a in g])
    assert np.array_equal(got, want)


def test_integer_bin_rule_equals_atan2_bins_exhaustively():
    """The round-5 HIP kernel decides the bin without floating point: (g, c) = gradient flipped into the upper half plane,
    cnt = #{j in 1..4 : |c| <= (g * round(cot(20 j deg) * 2^16)) >> 16}, bin = cnt if c >= 0 else 8 - cnt (csrc/hog.hip
    hog_bin).  Same constants here, every integer gradient pair, against skimage's atan2 interval test."""
    Q = [180059, 78103, 37837, 11556]
    assert Q == [int(round(2 ** 16 / np.tan(np.deg2rad(a)))) for a in (20, 40, 60, 80)]
    g = np.arange(-255, 256)
    GR, GC = np.meshgrid(g, g, indexing='ij')
    ori = np.rad2deg(np.arctan2(GR.astype(np.float64), GC.astype(np.float64))) % 180
    want = np.zeros_like(GR)
    for i in range(9):
        want[(ori >= 20 * i) & (ori < 20 * (i + 1))] = i
    flip = (GR < 0) | ((GR == 0) & (GC < 0))
    gg, cc = np.where(flip, -GR, GR), np.where(flip, -GC, GC)
    a = np.abs(cc)
    cnt = sum((a <= ((gg * q) >> 16)).astype(np.int64) for q in Q)
    got = np.where(cc >= 0, cnt, 8 - cnt)
    got[(GR == 0) & (GC == 0)] = 0
    assert np.array_equal(got, want)


def test_edge_frames():
    for f in (np.zeros((224, 224, 3), np.uint8), np.full((224, 224, 3), 255, np.uint8)):
        assert np.array_equal(H.extract_hog_features(f), np.zeros((14, 14, 108)))


def test_hog_table_blob_encodes_host_hypot_as_rounded_sqrt_plus_correction():
    """vtx_hog_build_table (host code of the C-ABI library, no GPU): the blob is the 65 536 doubles hypot(c, r) of the host's
    libm followed by 4096 words with 2 bits per gradient pair = hypot minus the correctly rounded sqrt(r^2 + c^2) in ulps.
    The round-5 kernel computes the square root and applies that correction; here: decoding the words on the CPU gives
    back the hypot table bit for bit, and the corrections are what numpy sees (np.hypot vs np.sqrt of the exact integer)."""
    import ctypes
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'videotransformer-pytorch_amd'))
    from vtx import _lib
    lib = _lib.load()
    nbytes = lib.vtx_hog_table_bytes()
    assert nbytes == 65536 * 8 + 4096 * 4
    blob = np.zeros(nbytes // 8, dtype=np.float64)
    assert lib.vtx_hog_build_table(ctypes.c_void_p(blob.ctypes.data)) == 0
    tab = blob[:65536].reshape(256, 256)
    words = blob[65536:].view(np.uint32)
    r = np.arange(256, dtype=np.float64)
    assert np.array_equal(tab, np.hypot(r[None, :], r[:, None]))
    idx = np.arange(65536)
    code = (words[idx >> 4] >> (2 * (idx & 15))) & 3
    delta = np.where(code == 3, -1, code).astype(np.int64)
    root = np.sqrt(r[None, :] ** 2 + r[:, None] ** 2).reshape(-1)
    rebuilt = (root.view(np.int64) + delta).view(np.float64)
    assert np.array_equal(rebuilt, tab.reshape(-1))
    assert set(np.unique(code)) <= {0, 1, 3}
