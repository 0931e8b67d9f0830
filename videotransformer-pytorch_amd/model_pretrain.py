"""Drop-in for the reference ``model_pretrain.py``: the same command-line flags (reference :21-152) and the same
``single_run()`` (:154-230) -- linear learning-rate scaling, experiment tag, checkpoint / log directories, seeds,
``model_trainer.VideoTransformer`` -- with the data-parallel training loop built on this package instead of on
``pl.Trainer(accelerator="ddp", precision=16, plugins=[DDPPlugin(...)])`` (:200-211):

  * one process per GPU: started under ``torch.distributed.run`` it reads RANK / LOCAL_RANK / WORLD_SIZE; started
    plainly with more than one GPU selected (``-gpus``) it re-launches itself with one rank per GPU on 127.0.0.1;
  * rank 0's parameters are broadcast once (``vtx.dp.broadcast_parameters``: what DDP does at wrap time), gradients live
    in the flat per-layer buckets of ``vtx.dp.GradBuckets(direct=True)`` and are mean-all-reduced over RCCL / xGMI
    bucket by bucket while backward is still running (what DDP's reducer does);
  * every step runs the LightningModule's own hooks in Lightning's order -- ``training_step`` -> ``backward`` ->
    ``on_after_backward`` -> ``optimizer_step`` -- under ``torch.autocast('cuda', bfloat16)`` (the AMP class of
    ``precision=16``; bf16 needs no GradScaler), then per epoch the LR scheduler, ``training_epoch_end``, validation and
    test epochs when their data is given; checkpoints hold model, optimizer, scheduler and epoch and resume through
    ``-resume`` / ``-resume_from_checkpoint``.

The reference's data pipeline (``data_trainer.KineticsDataModule``: decord, kornia, OpenCV on CPU workers) is outside
this package's scope; when those modules are importable next to this file they are used unchanged (with a
DistributedSampler per rank, as Lightning would inject), otherwise ``-synthetic_steps N`` runs N steps per epoch on
synthetic clips of the configured shape (supervised: clips + labels; mim: clips + on-device HOG targets + cube masks),
which is how the tests and benchmarks drive this entry point.
"""
import argparse
import os
import random
import subprocess
import sys
import time
import warnings

import numpy as np
import torch
import torch.distributed as dist

from utils import print_on_rank_zero


def build_parser():
    """The reference's flags, names / types / defaults / required as at model_pretrain.py:21-152 (including its
    ``type=bool`` flags, for which any non-empty value is True), plus ``-synthetic_steps``."""
    parser = argparse.ArgumentParser(description='lr receiver')
    add = parser.add_argument
    # Common
    add('-epoch', type=int, required=True, help='the max epochs of training')
    add('-batch_size', type=int, required=True, help='the batch size of data inputs')
    add('-num_workers', type=int, default=4, help='the num workers of loading data')
    add('-resume', default=False, action='store_true')
    add('-resume_from_checkpoint', type=str, default=None, help='the pretrain params from specific path')
    add('-log_interval', type=int, default=30, help='the intervals of logging')
    add('-save_ckpt_freq', type=int, default=20, help='the intervals of saving model')
    add('-objective', type=str, default='mim', help='the learning objective from [mim, supervised]')
    add('-eval_metrics', type=str, default='finetune', help='the eval metrics choosen from [linear_prob, finetune]')
    # Environment
    add('-gpus', nargs='+', type=int, default=-1, help='the avaiable gpus in this experiment')
    add('-root_dir', type=str, required=True, help='the path to root dir for work space')
    # Data
    add('-num_class', type=int, required=True, help='the num class of dataset used')
    add('-num_samples_per_cls', type=int, default=10000, help='the num samples of per class')
    add('-img_size', type=int, default=224, help='the size of processed image')
    add('-num_frames', type=int, required=True, help='the mumber of frame sampling')
    add('-frame_interval', type=int, required=True, help='the intervals of frame sampling')
    add('-data_statics', type=str, default='kinetics', help='choose data statics from [imagenet, kinetics]')
    add('-train_data_path', type=str, required=True, help='the path to train set')
    add('-val_data_path', type=str, default=None, help='the path to val set')
    add('-test_data_path', type=str, default=None, help='the path to test set')
    add('-multi_crop', type=bool, default=False, help='Whether or not to use multi crop.')
    add('-mixup', type=bool, default=False, help='Whether or not to use mixup / cutmix.')
    add('-auto_augment', type=str, default=None, help='the used Autoaugment policy')
    # Model
    add('-arch', type=str, default='timesformer', help='the choosen model arch from [timesformer, vivit]')
    add('-attention_type', type=str, default='divided_space_time', help='the choosen attention type using in model')
    add('-pretrain_pth', type=str, default=None, help='the path to the pretrain weights')
    add('-weights_from', type=str, default='imagenet', help='the pretrain params from [imagenet, kinetics]')
    # Training / optimization
    add('-seed', type=int, default=0, help='the seed of exp')
    add('-optim_type', type=str, default='adamw', help='the optimizer using in the training')
    add('-lr_schedule', type=str, default='cosine', help='the lr schedule using in the training')
    add('-lr', type=float, required=True, help='the initial learning rate')
    add('-layer_decay', type=float, default=0.75, help='the value of layer_decay')
    add('--min_lr', type=float, default=1e-6, help='Target LR at the end of optimization (cosine schedule with linear warmup).')
    add('-use_fp16', type=bool, default=True, help='Mixed precision (here: bf16 storage with fp32 accumulation).')
    add('-weight_decay', type=float, default=0.05, help='Initial value of the weight decay.')
    add('-weight_decay_end', type=float, default=0.05, help='Final value of the weight decay (cosine schedule).')
    add('-clip_grad', type=float, default=0, help='Maximal parameter gradient norm if using gradient clipping. 0 for disabling.')
    add('-warmup_epochs', default=5, type=int, help='Number of epochs for the linear learning-rate warm up.')
    # this package only
    add('-synthetic_steps', type=int, default=0,
        help='> 0: train on synthetic clips for this many steps per epoch instead of the reference data pipeline')
    return parser


def parse_args(argv=None):
    return build_parser().parse_args(argv)


def experiment_tag(args):
    """The results directory name of the reference (model_pretrain.py:168-176).  The reference's string is ~300 characters
    long -- longer than the 255-byte file-name limit of ext4 / xfs / overlayfs, where its ``os.makedirs`` raises
    ``OSError: File name too long`` -- so a tag over the limit is cut to its first 200 characters + '_' + 12 hex digits of
    its SHA-1 (still one directory per flag combination, still recognisable)."""
    tag = _reference_tag(args)
    if len(tag.encode()) > 255:
        import hashlib
        tag = tag[:200] + '_' + hashlib.sha1(tag.encode()).hexdigest()[:12]
    return tag


def _reference_tag(args):
    return (f'objective_{args.objective}_arch_{args.arch}_lr_{args.lr}_'
            f'optim_{args.optim_type}_lr_schedule_{args.lr_schedule}_'
            f'fp16_{args.use_fp16}_weight_decay_{args.weight_decay}_'
            f'weight_decay_end_{args.weight_decay_end}_warmup_epochs_{args.warmup_epochs}_'
            f'pretrain_{args.pretrain_pth}_weights_from_{args.weights_from}_seed_{args.seed}_'
            f'img_size_{args.img_size}_num_frames_{args.num_frames}_eval_metrics_{args.eval_metrics}_'
            f'frame_interval_{args.frame_interval}_mixup_{args.mixup}_'
            f'multi_crop_{args.multi_crop}_auto_augment_{args.auto_augment}_')


def selected_gpus(args):
    """Device indices the run uses: ``-gpus 0 1 2 3`` (a list) or every visible GPU (the default -1), as the reference
    counts them for its learning-rate scaling (model_pretrain.py:159-163)."""
    if isinstance(args.gpus, int):
        return list(range(torch.cuda.device_count()))
    return list(args.gpus)


# ------------------------------------------------------------------------------------------------ synthetic data
class SyntheticBatches:
    """``steps`` batches per epoch of the configured shape, already on the device, as the reference's collate function
    delivers them (data_trainer.py:10-36): supervised ``[clips [B,T,3,H,W] fp32, labels [B] int64]``; mim ``[clips,
    HOG targets [B,T,14,14,108] f64, mask [B,T/2,14,14], cube markers]`` with the targets of the masked cubes' centre
    frames computed by ``vtx_hog_fwd`` from uint8 frames (dataset.py:188-196 does that on CPU workers)."""

    def __init__(self, args, steps, device, rank=0):
        self.args, self.steps, self.device = args, int(steps), device
        self.gen = torch.Generator().manual_seed(1234 + 7919 * rank + args.seed)

    def __len__(self):
        return self.steps

    def __iter__(self):
        from vtx import ops
        a, dev = self.args, self.device
        B = a.batch_size
        for _ in range(self.steps):
            if a.objective == 'mim':
                T, S = 16, 224                                   # MaskFeat() as model_trainer builds it (:53-54)
                clips = torch.randn(B, T, 3, S, S, generator=self.gen).to(dev)
                mask = torch.zeros(B, T // 2, 14, 14, dtype=torch.int32)
                markers = []
                target = torch.zeros(B, T, 14, 14, 108, dtype=torch.float64, device=dev)
                for b in range(B):
                    start = int(torch.randint(0, T // 2 - 2, (1,), generator=self.gen))
                    span = int(torch.randint(1, 3, (1,), generator=self.gen))
                    y0, x0 = (int(v) for v in torch.randint(0, 7, (2,), generator=self.gen))
                    mask[b, start:start + span, y0:y0 + 7, x0:x0 + 7] = 1
                    markers.append([[start, span]])
                    frame = torch.randint(0, 256, (1, S, S, 3), generator=self.gen, dtype=torch.uint8).to(dev)
                    target[b, start * 2 + span * 2 // 2] = ops.hog_fwd(frame)[0]
                yield [clips, target, mask.to(dev), markers]
            else:
                clips = torch.randn(B, a.num_frames, 3, a.img_size, a.img_size, generator=self.gen).to(dev)
                labels = torch.randint(0, a.num_class, (B,), generator=self.gen).to(dev)
                yield [clips, labels]


def _reference_data_module(args):
    """The reference's LightningDataModule when its data files sit next to this one (out of this package's scope)."""
    try:
        from data_trainer import KineticsDataModule
    except ImportError as e:
        raise SystemExit('model_pretrain: the reference data pipeline (data_trainer.py / dataset.py / data_transform.py and their '
                         f'decord / kornia / cv2 dependencies) is not importable here ({e}); pass -synthetic_steps N to run on '
                         'synthetic clips') from e
    return KineticsDataModule(configs=args, train_ann_path=args.train_data_path, val_ann_path=args.val_data_path,
                              test_ann_path=args.test_data_path)


def _sharded(loader, rank, world, shuffle, epoch):
    """A DataLoader over this rank's shard (what Lightning's DDP plugin does by injecting a DistributedSampler)."""
    if world == 1 or loader is None:
        return loader
    from torch.utils.data import DataLoader
    from torch.utils.data.distributed import DistributedSampler
    sampler = DistributedSampler(loader.dataset, num_replicas=world, rank=rank, shuffle=shuffle, drop_last=loader.drop_last)
    sampler.set_epoch(epoch)
    return DataLoader(loader.dataset, batch_size=loader.batch_size, num_workers=loader.num_workers, collate_fn=loader.collate_fn,
                      sampler=sampler, drop_last=loader.drop_last, pin_memory=loader.pin_memory)


def _to_device(batch, device):
    return [t.to(device, non_blocking=True) if torch.is_tensor(t) else t for t in batch]


# ------------------------------------------------------------------------------------------------ the trainer
class Trainer:
    """The members of ``pl.Trainer`` that ``model_trainer.VideoTransformer`` touches (``max_epochs``, ``current_epoch``,
    ``save_checkpoint``) and ``fit``: the native data-parallel loop described in the module docstring."""

    def __init__(self, args, device, rank=0, world=1, force_comm=False):
        self.args, self.device, self.rank, self.world = args, device, rank, world
        self.max_epochs = args.epoch
        self.current_epoch = 0
        self.global_step = 0
        self.force_comm = force_comm
        self.model = self.optimizer = self.scheduler = self.buckets = None
        self.last_loss = None

    # -- checkpoints (rank 0 writes; every rank reads) --------------------------------------------
    def save_checkpoint(self, path):
        if self.rank != 0:
            return
        # Lightning 1.3.8 (the reference's pin) stores current_epoch + 1 and resumes at that value: same convention here, so a
        # reference-written last_checkpoint.pth neither skips nor repeats an epoch ('callbacks' is not written: no callback state)
        state = {'epoch': self.current_epoch + 1, 'global_step': self.global_step, 'state_dict': self.model.state_dict(),
                 'pytorch-lightning_version': '1.3.8',
                 'optimizer_states': [self.optimizer.state_dict()],
                 'lr_schedulers': [self.scheduler.state_dict() if self.scheduler is not None else None]}
        tmp = path + '.tmp'
        torch.save(state, tmp)
        os.replace(tmp, path)

    def _resume(self, path):
        state = torch.load(path, map_location=self.device)
        self.model.load_state_dict(state['state_dict'])
        self.optimizer.load_state_dict(state['optimizer_states'][0])
        if self.scheduler is not None and state['lr_schedulers'][0] is not None:
            self.scheduler.load_state_dict(state['lr_schedulers'][0])
        self.current_epoch = int(state['epoch'])
        self.global_step = int(state.get('global_step', 0))
        from vtx import functions
        functions.clear_weight_cache()
        print_on_rank_zero(f'resumed from {path}: continuing at epoch {self.current_epoch}')

    # -- one optimisation step, in Lightning's hook order ------------------------------------------
    def _train_step(self, model, batch, batch_idx):
        self.buckets.zero()
        with torch.autocast('cuda', dtype=torch.bfloat16, enabled=bool(self.args.use_fp16)):
            out = model.training_step(batch, batch_idx)
        loss = out['loss']
        loss.backward()
        self.buckets.finish()                                   # every bucket's mean all-reduce has landed in .grad
        if hasattr(self.optimizer, 'set_skipped'):              # parameters outside this step's graph take no update (no decay)
            self.optimizer.set_skipped(self.buckets.unfired())
        model.on_after_backward()
        model.optimizer_step(self.current_epoch, batch_idx, self.optimizer, 0, None, False, True, False)
        self.global_step += 1
        self.last_loss = loss.detach()
        return out

    def _eval_epoch(self, model, loader, step, end):
        if loader is None:
            return
        model.eval()
        with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16, enabled=bool(self.args.use_fp16)):
            for i, batch in enumerate(loader):
                step(_to_device(batch, self.device), i)
        end([])
        model.train()

    def fit(self, model, data):
        from vtx import dp
        args = self.args
        self.model = model.to(self.device)
        model.train()
        dp.broadcast_parameters(model)
        (optimizers, schedulers) = model.configure_optimizers()
        self.optimizer, self.scheduler = optimizers[0], schedulers[0]
        model._optimizers = self.optimizer                      # what self.optimizers() hands to the hooks without Lightning
        # gradient buckets over the parameters the optimizer updates
        updated = {id(p) for g in self.optimizer.param_groups for p in g['params']}
        params = [p for p in model.parameters() if id(p) in updated]      # registration order: buckets fill in backward order
        self.buckets = dp.GradBuckets(params, force_comm=self.force_comm, direct=True)
        if args.resume_from_checkpoint:
            if os.path.isfile(args.resume_from_checkpoint):
                self._resume(args.resume_from_checkpoint)
            else:
                print_on_rank_zero(f'no checkpoint at {args.resume_from_checkpoint}: starting from scratch')
        synthetic = isinstance(data, SyntheticBatches)
        if not synthetic:
            data.setup('fit')
        try:
            while self.current_epoch < self.max_epochs:
                ep = self.current_epoch
                loader = data if synthetic else _sharded(data.train_dataloader(), self.rank, self.world, True, ep)
                outputs = []
                t0 = time.perf_counter()
                model.data_start = time.perf_counter()
                for i, batch in enumerate(loader):
                    out = self._train_step(model, _to_device(batch, self.device), i)
                    outputs.append({'data_time': out['data_time']})
                    if (i + 1) % max(args.log_interval, 1) == 0:
                        print_on_rank_zero(f'epoch {ep} step {i + 1}/{len(loader)} loss {float(self.last_loss):.4f} '
                                           f'lr {self.optimizer.param_groups[0]["lr"]:.3e}')
                torch.cuda.synchronize(self.device)
                n = max(len(outputs), 1)
                print_on_rank_zero(f'epoch {ep}: {n} steps, {(time.perf_counter() - t0) / n * 1e3:.1f} ms/step on {self.world} GPU(s), '
                                   f'loss {float(self.last_loss) if self.last_loss is not None else float("nan"):.4f}')
                if self.scheduler is not None:
                    self.scheduler.step()                        # epoch-wise, as Lightning steps the reference's schedulers
                model.training_epoch_end(outputs)
                if not synthetic:
                    self._eval_epoch(model, _sharded(data.val_dataloader(), self.rank, self.world, False, ep), model.validation_step,
                                     model.validation_epoch_end)
                self.current_epoch += 1
            if not synthetic and model.do_test:
                data.setup('test')
                self._eval_epoch(model, _sharded(data.test_dataloader(), self.rank, self.world, False, 0), model.test_step,
                                 model.test_epoch_end)
        finally:
            self.buckets.remove()
        return model


# ------------------------------------------------------------------------------------------------ entry point
def _free_port():
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _spawn_ranks(gpus, argv):
    """Plain ``python model_pretrain.py ... -gpus 0 1 2 3``: one rank per selected GPU under torch.distributed.run."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'),
               VTX_PRETRAIN_GPUS=','.join(str(g) for g in gpus))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={len(gpus)}', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.abspath(__file__)] + list(argv)
    return subprocess.run(cmd, env=env).returncode


def single_run(argv=None):
    args = parse_args(argv)
    warnings.filterwarnings('ignore')
    if not torch.cuda.is_available():
        raise SystemExit('model_pretrain: needs a GPU (this package has no CPU path)')
    gpus = selected_gpus(args)
    num_gpus = len(gpus)
    under_launcher = 'WORLD_SIZE' in os.environ and 'RANK' in os.environ
    if num_gpus > 1 and not under_launcher:
        sys.exit(_spawn_ranks(gpus, sys.argv[1:] if argv is None else argv))
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    mine = [int(g) for g in os.environ['VTX_PRETRAIN_GPUS'].split(',')] if os.environ.get('VTX_PRETRAIN_GPUS') else gpus
    device = torch.device('cuda', mine[local] if local < len(mine) else local)
    torch.cuda.set_device(device)
    force_comm = os.environ.get('VTX_FORCE_DP', '0') == '1'     # a 1-rank RCCL group: the whole exchange path on one GPU
    if (world > 1 or force_comm) and not dist.is_initialized():
        from vtx import dp as _dp
        _dp.init_process_group(device, rank, world)

    # linear learning rate scale (model_pretrain.py:158-164): per-GPU batch x number of GPUs / 256
    effective_batch_size = args.batch_size * (world if under_launcher else max(num_gpus, 1))
    args.lr = args.lr * effective_batch_size / 256

    exp_tag = experiment_tag(args)
    ckpt_dir = os.path.join(args.root_dir, f'results/{exp_tag}/ckpt')
    log_dir = os.path.join(args.root_dir, f'results/{exp_tag}/log')
    os.makedirs(ckpt_dir, exist_ok=True)
    os.makedirs(log_dir, exist_ok=True)
    do_eval = args.val_data_path is not None
    do_test = args.test_data_path is not None
    if args.resume and not args.resume_from_checkpoint:
        args.resume_from_checkpoint = os.path.join(ckpt_dir, 'last_checkpoint.pth')

    import vtx
    vtx.set_precision('auto')                                   # bf16 inside the autocast region of the loop, fp32 outside
    trainer = Trainer(args, device, rank=rank, world=world, force_comm=force_comm)

    # To be reproducable (model_pretrain.py:213-217)
    torch.random.manual_seed(args.seed)
    np.random.seed(args.seed)
    random.seed(args.seed)

    from model_trainer import VideoTransformer
    model = VideoTransformer(configs=args, trainer=trainer, ckpt_dir=ckpt_dir, do_eval=do_eval and not args.synthetic_steps,
                             do_test=do_test and not args.synthetic_steps)
    print_on_rank_zero(args)
    print_on_rank_zero(f'{time.strftime("%Y-%m-%d %H:%M:%S", time.localtime())} - INFO - Start running,')
    data = SyntheticBatches(args, args.synthetic_steps, device, rank) if args.synthetic_steps > 0 else _reference_data_module(args)
    try:
        trainer.fit(model, data)
    except BaseException:
        # no barrier on the error path: the peers sit in a gradient all-reduce, a barrier here would hang until the backend's
        # timeout and hide the exception -- tear the group down and let the launcher end the other ranks
        if dist.is_initialized():
            dist.destroy_process_group()
        raise
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    return trainer


if __name__ == '__main__':
    single_run()
