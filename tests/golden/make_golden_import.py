"""Golden key maps of the pretrained-checkpoint importers, produced by the UNMODIFIED reference
functions (weight_init.py:107-315) on small synthetic checkpoints (dev container only):

    python tests/golden/make_golden_import.py

For every (importer, attention_type, conv_type, copy / extend strategy) the reference function is run
against a module whose load_state_dict records what it is given; stored: key -> [shape, sum, abs-sum].
"""
import json
import os
import sys
import tempfile

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import ref_loader  # noqa: E402
from import_cases import CASES, synth_checkpoint, summarize  # noqa: E402


class Recorder:
    def load_state_dict(self, sd, strict=True):
        self.sd = dict(sd)
        return [], []


def main():
    R = ref_loader.load()
    W = R.weight_init
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for name, kind, kw in CASES:
            path = os.path.join(tmp, name + '.pth')
            torch.save(synth_checkpoint(kind), path)
            rec = Recorder()
            if kind == 'vit':
                W.init_from_vit_pretrain_(rec, path, **kw)
            elif kind == 'mae':
                W.init_from_mae_pretrain_(rec, path, **kw)
            else:
                W.init_from_kinetics_pretrain_(rec, path)
            out[name] = summarize(rec.sd)
            print(name, len(out[name]))
    json.dump(out, open(os.path.join(HERE, 'weight_import.json'), 'w'), indent=0, sort_keys=True)


if __name__ == '__main__':
    main()
