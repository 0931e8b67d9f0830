"""Golden vectors for TimeSformer-B at 16 frames (BASELINE.json north_star: 16x3x224x224 clips), produced by
the REFERENCE (through oracle/ref_loader.py) in the dev container; same conventions as make_golden.py.

    python tests/golden/make_golden_t16.py        # writes tests/golden/tsf_b_t16_train.npz
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from oracle import ref_loader, synth           # noqa: E402
from make_golden import run_model               # noqa: E402


def main():
    R = ref_loader.load()
    VT = R.video_transformer
    torch.set_num_threads(8)
    m = VT.TimeSformer(num_frames=16)
    sd = synth.synth_state_dict(synth.shapes_of(m), seed=0)
    r = run_model(m, synth.synth_clip(1, 16, seed=21), sd, train=True, seed=9)
    np.savez_compressed(os.path.join(HERE, 'tsf_b_t16_train.npz'), **r)
    print('tsf_b_t16_train', len(r), r['out'].shape, float(np.abs(r['out']).max()))


if __name__ == '__main__':
    main()
