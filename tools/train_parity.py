"""Do the bf16 kernels TRAIN like the reference?  (GPU box; the oracle is the checker.)

A small TimeSformer (divided_space_time, D 128, `layers` layers, 4 frames of 64 x 64) takes K SGD(nesterov) steps on a fixed batch
with a bounded loss (0.5 |y - target|^2) and DropPath on (the same CPU draws in every arm), in four arms from the same start:

    fp32 oracle          oracle/vt_oracle.py on the CPU in float32 (the reference's arithmetic)            <- the yardstick
    oracle + autocast    the same graph under torch.autocast('cpu', bfloat16): what the reference's AMP does
    vtx bf16             this library, bf16 kernels, bf16 residual stream (the default, the benchmarked mode)
    vtx bf16 exact       this library, bf16 kernels, vtx.set_stream('fp32')
    vtx exact + grad     ... vtx.set_stream('fp32+grad'): the stream's gradient in float32 too

and prints, per arm, the loss at every step and -- after K steps -- the relative L2 distance of ALL parameters from the fp32 oracle's
(the drift of the training trajectory) and of the model output on a held-out clip.

    python tools/train_parity.py [--steps 12] [--layers 6] [--lr 0.02] [--seeds 3]
    python tools/train_parity.py --full 1 [--steps 6] [--batch 4] [--seeds 1]      # TimeSformer-B itself: D 768, 12 layers, 8 x 224^2 clips
    python tools/train_parity.py --model vivit [--full 1]                          # ViViT fact_encoder, Conv3d tubelets (full: ViViT-B, 16 x 224^2)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'videotransformer-pytorch_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402

DEV = 'cuda:0'


def main():
    opt = lambda name, d: type(d)(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else d      # noqa: E731
    K, L, lr, nseeds = opt('--steps', 12), opt('--layers', 6), opt('--lr', 0.02), opt('--seeds', 3)
    import vtx
    import video_transformer as V
    from vtx import optim
    from oracle import synth, vt_oracle as O
    torch.set_num_threads(min(os.cpu_count() or 8, 16))
    for kv in [a.split('=', 1) for i, a in enumerate(sys.argv) if i and sys.argv[i - 1] == '--opt']:      # --opt attn_valu=1 (vtx.set_option)
        vtx.set_option(kv[0], kv[1])
        print(f'option {kv[0]} = {kv[1]}')
    full = opt('--full', 0)
    top = opt('--top', 0)
    which = opt('--model', 'timesformer')
    vivit = which == 'vivit'
    if full:
        L, T, S, D, H = 12, (16 if vivit else 8), 224, 768, 12
    else:
        T, S, D, H = (8 if vivit else 4), 64, 128, 2
    cfg = dict(num_frames=T, img_size=S, patch_size=16, embed_dims=D, num_heads=H, num_transformer_layers=L)
    Model = V.ViViT if vivit else V.TimeSformer
    if vivit:
        cfg.update(attention_type='fact_encoder', conv_type='Conv3d', tube_size=2)
    fwd = ((lambda ps, xx, training=False: O.vivit_forward(ps, xx, T, tube_size=2, heads=H, layers=L, training=training)) if vivit else
           (lambda ps, xx, training=False: O.timesformer_forward(ps, xx, T, heads=H, layers=L, training=training)))
    B = opt('--batch', 4)
    print(f'{"ViViT fact_encoder (Conv3d tubelet 2)" if vivit else "TimeSformer divided_space_time"} D {D}, {L} layers, {T} x {S}^2 clips, batch {B}, {K} SGD(nesterov, lr {lr}, momentum 0.9) steps, DropPath 0.1, '
          f'loss 0.5 |y - t|^2; distances are relative L2 over ALL parameters / the held-out output, against the fp32 oracle arm', flush=True)
    tot = {}
    for seed in range(nseeds):
        shapes = synth.shapes_of(Model(**cfg))
        sd0 = synth.synth_state_dict(shapes, seed)
        x = synth.synth_clip(B, T, 3, S, S, seed=10 + seed)
        xh = synth.synth_clip(2, T, 3, S, S, seed=50 + seed)
        tgt = synth.synth_tensor('target', (B, D), seed) * 0.5

        def run_oracle(autocast):
            ps = {k: v.clone().requires_grad_(True) for k, v in sd0.items()}
            o = torch.optim.SGD(list(ps.values()), lr=lr, momentum=0.9, nesterov=True)
            losses = []
            for step in range(K):
                o.zero_grad()
                torch.manual_seed(1000 + step)
                with torch.autocast('cpu', dtype=torch.bfloat16, enabled=autocast):
                    y = fwd(ps, x, training=True)
                loss = 0.5 * ((y.float() - tgt) ** 2).sum()
                loss.backward()
                o.step()
                losses.append(loss.item())
            with torch.no_grad(), torch.autocast('cpu', dtype=torch.bfloat16, enabled=autocast):
                yh = fwd(ps, xh).float()
            return {k: v.detach() for k, v in ps.items()}, yh, losses

        def run_vtx(stream, precision='bf16'):
            vtx.set_precision(precision)
            vtx.set_stream(stream)
            m = Model(**cfg)
            m.load_state_dict(sd0, strict=True)
            m.to(DEV).train()
            o = optim.FusedSGD(m.parameters(), lr=lr, momentum=0.9, nesterov=True)
            losses = []
            for step in range(K):
                m.zero_grad(set_to_none=True)
                torch.manual_seed(1000 + step)
                y = m(x.to(DEV))
                loss = 0.5 * ((y - tgt.to(DEV)) ** 2).sum()
                loss.backward()
                o.step()
                losses.append(loss.item())
            m.eval()
            with torch.no_grad():
                yh = m(xh.to(DEV)).float().cpu()
            vtx.set_stream('bf16')
            return {k: v.detach().cpu() for k, v in m.state_dict().items()}, yh, losses

        ref_p, ref_y, ref_l = run_oracle(False)
        arms = {'oracle + autocast': run_oracle(True), 'vtx bf16': run_vtx('bf16'), 'vtx bf16 exact': run_vtx('fp32'),
                'vtx exact + grad': run_vtx('fp32+grad')}
        if opt('--fp32-arm', 0):
            arms['vtx fp32 kernels'] = run_vtx('bf16', 'fp32')          # validates the harness: the same trajectory to float32 rounding
        flat = lambda p: torch.cat([p[k].double().flatten() for k in sorted(ref_p)])      # noqa: E731
        fr = flat(ref_p)
        moved = (fr - flat(sd0)).norm().item() / fr.norm().item()
        print(f'seed {seed}: fp32 oracle loss {ref_l[0]:.4f} -> {ref_l[-1]:.4f}; the parameters moved by {moved:.3e} (relative L2) in {K} steps')
        for name, (p, yh, ls) in arms.items():
            dp = (flat(p) - fr).norm().item() / fr.norm().item()
            dstep = (flat(p) - fr).norm().item() / (fr - flat(sd0)).norm().item()
            dy = (yh.double() - ref_y.double()).norm().item() / ref_y.double().norm().item()
            dl = max(abs(a - b) / max(abs(b), 1e-30) for a, b in zip(ls, ref_l))
            print(f'   {name:18s} parameters {dp:.3e} of |w| = {dstep:.3e} of the distance travelled; held-out output {dy:.3e}; worst loss deviation {dl:.3e}; '
                  f'final loss {ls[-1]:.4f}')
            if top:
                tot2 = (flat(p) - fr).norm().item() ** 2
                per = sorted(((((p[k].double() - ref_p[k].double()).norm().item() ** 2) / tot2, k) for k in ref_p), reverse=True)[:top]
                own = lambda q, k: (q[k].double() - ref_p[k].double()).norm().item() / max((ref_p[k].double() - sd0[k].double()).norm().item(), 1e-30)      # noqa: E731
                amp = arms['oracle + autocast'][0]
                print('      largest shares of the squared drift (tensor: share, error relative to its own travel, the same for the reference AMP arm):')
                for sh, k in per:
                    print(f'         {k}: {sh:.3f}  {own(p, k):.3e}  (AMP {own(amp, k):.3e}: x {own(p, k) / max(own(amp, k), 1e-30):.2f})')
            t = tot.setdefault(name, [0.0, 0.0, 0.0])
            t[0] += dstep / nseeds; t[1] += dy / nseeds; t[2] += dl / nseeds
    print('means over the seeds:')
    for name, t in tot.items():
        print(f'   {name:18s} trajectory drift {t[0]:.3e} of the distance travelled; held-out output {t[1]:.3e}; worst loss deviation {t[2]:.3e}')


if __name__ == '__main__':
    main()
