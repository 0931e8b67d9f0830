"""The drop-in ``model_pretrain.py`` keeps the reference's command line (reference model_pretrain.py:21-152): every flag with
the same option string, type, default, ``required`` and ``nargs`` / ``action`` -- read from the reference's source by AST
when /root/reference is present, and from the list frozen below (extracted the same way) on any host."""
import ast
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference/model_pretrain.py'
REQUIRED = ['-epoch', '1', '-batch_size', '2', '-root_dir', '/tmp/vtx_pretrain_flags', '-num_class', '10', '-num_frames', '8',
            '-frame_interval', '4', '-train_data_path', '/nowhere/train.txt', '-lr', '0.001']

# (option, type name, default, required, nargs, action) of every add_argument call of the reference
FROZEN = [
    ('-epoch', 'int', None, True, None, None), ('-batch_size', 'int', None, True, None, None), ('-num_workers', 'int', 4, False, None, None),
    ('-resume', None, False, False, None, 'store_true'), ('-resume_from_checkpoint', 'str', None, False, None, None),
    ('-log_interval', 'int', 30, False, None, None), ('-save_ckpt_freq', 'int', 20, False, None, None),
    ('-objective', 'str', 'mim', False, None, None), ('-eval_metrics', 'str', 'finetune', False, None, None),
    ('-gpus', 'int', -1, False, '+', None), ('-root_dir', 'str', None, True, None, None), ('-num_class', 'int', None, True, None, None),
    ('-num_samples_per_cls', 'int', 10000, False, None, None), ('-img_size', 'int', 224, False, None, None),
    ('-num_frames', 'int', None, True, None, None), ('-frame_interval', 'int', None, True, None, None),
    ('-data_statics', 'str', 'kinetics', False, None, None), ('-train_data_path', 'str', None, True, None, None),
    ('-val_data_path', 'str', None, False, None, None), ('-test_data_path', 'str', None, False, None, None),
    ('-multi_crop', 'bool', False, False, None, None), ('-mixup', 'bool', False, False, None, None),
    ('-auto_augment', 'str', None, False, None, None), ('-arch', 'str', 'timesformer', False, None, None),
    ('-attention_type', 'str', 'divided_space_time', False, None, None), ('-pretrain_pth', 'str', None, False, None, None),
    ('-weights_from', 'str', 'imagenet', False, None, None), ('-seed', 'int', 0, False, None, None),
    ('-optim_type', 'str', 'adamw', False, None, None), ('-lr_schedule', 'str', 'cosine', False, None, None),
    ('-lr', 'float', None, True, None, None), ('-layer_decay', 'float', 0.75, False, None, None), ('--min_lr', 'float', 1e-06, False, None, None),
    ('-use_fp16', 'bool', True, False, None, None), ('-weight_decay', 'float', 0.05, False, None, None),
    ('-weight_decay_end', 'float', 0.05, False, None, None), ('-clip_grad', 'float', 0, False, None, None),
    ('-warmup_epochs', 'int', 5, False, None, None),
]


def _flags_of(path):
    out = []
    for node in ast.walk(ast.parse(open(path).read())):
        if isinstance(node, ast.Call) and getattr(node.func, 'attr', None) == 'add_argument':
            kw = {k.arg: k.value for k in node.keywords}
            lit = lambda k: ast.literal_eval(kw[k]) if k in kw else None       # noqa: E731
            out.append((ast.literal_eval(node.args[0]), kw['type'].id if 'type' in kw else None, lit('default'),
                        bool(lit('required')), lit('nargs'), lit('action')))
    return out


@pytest.mark.skipif(not os.path.isfile(REF), reason='reference checkout not on this host')
def test_frozen_flag_list_is_the_reference_source():
    assert _flags_of(REF) == FROZEN


def test_drop_in_parser_has_every_reference_flag():
    sys.path.insert(0, os.path.join(ROOT, 'videotransformer-pytorch_amd'))
    import model_pretrain as MP
    acts = {a.option_strings[0]: a for a in MP.build_parser()._actions if a.option_strings}
    for opt, tname, default, required, nargs, action in FROZEN:
        a = acts[opt]
        assert a.required == required and a.nargs == (0 if action == 'store_true' else nargs), opt
        assert a.default == default and type(a.default) is type(default), (opt, a.default, default)
        assert (a.type.__name__ if a.type else None) == tname, opt
    assert set(acts) - {f[0] for f in FROZEN} == {'-h', '-synthetic_steps'}
    args = MP.parse_args(REQUIRED + ['-gpus', '0', '1', '-mixup', 'False', '-resume'])
    assert args.gpus == [0, 1] and args.mixup is True and args.resume is True      # type=bool: any non-empty string is True
    assert args.min_lr == 1e-6 and args.use_fp16 is True and args.objective == 'mim' and args.synthetic_steps == 0
    assert MP.selected_gpus(args) == [0, 1]
    tag = MP._reference_tag(args)
    short = MP.experiment_tag(args)
    assert len(short.encode()) <= 255 and short.startswith(tag[:200]) and MP.experiment_tag(args) == short
    assert tag.startswith('objective_mim_arch_timesformer_lr_0.001_optim_adamw_lr_schedule_cosine_fp16_True_weight_decay_0.05_')
    assert tag.endswith('frame_interval_4_mixup_True_multi_crop_False_auto_augment_None_')
    with pytest.raises(SystemExit):
        MP.parse_args(['-epoch', '1'])                       # required flags missing
