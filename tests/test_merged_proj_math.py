"""The algebra behind the merged attn.proj + temporal_fc GEMM (vtx.functions.TimeAttnFn), restated with
CPU torch ops step for step as the HIP path performs it (product weight, bias b_c + b_tfc / c under the row scale, fix-up
of dropped sequences, zeroed rows of the attention output, gradients mapped back to the two Linear layers) and checked
against autograd through the reference's own formulation: proj -> per-sequence DropPath -> temporal_fc -> residual
(reference transformer.py:268-275)."""
import torch


def _reference(o, x, s_rows, wp, bp, wt, bt):
    a = torch.nn.functional.linear(o, wp, bp) * s_rows[:, None]          # attn.proj, then DropPath (scale 0 or 1/keep)
    return x + torch.nn.functional.linear(a, wt, bt)                      # temporal_fc + residual


def test_merged_projection_matches_two_linears_and_their_gradients():
    torch.manual_seed(0)
    D, T, S = 24, 4, 10                      # S sequences of T rows
    M = S * T
    keep = 0.7
    c = 1.0 / keep
    s_seq = (torch.rand(S) < keep).double() * c
    assert 0 < (s_seq == 0).sum() < S
    s_rows = s_seq.repeat_interleave(T)
    o = torch.randn(M, D, dtype=torch.float64)
    x = torch.randn(M, D, dtype=torch.float64)
    params = [torch.randn(D, D, dtype=torch.float64, requires_grad=True), torch.randn(D, dtype=torch.float64, requires_grad=True),
              torch.randn(D, D, dtype=torch.float64, requires_grad=True), torch.randn(D, dtype=torch.float64, requires_grad=True)]
    wp, bp, wt, bt = params
    o_ref = o.clone().requires_grad_(True)
    out_ref = _reference(o_ref, x, s_rows, wp, bp, wt, bt)
    dout = torch.randn(M, D, dtype=torch.float64)
    out_ref.backward(dout)

    with torch.no_grad():
        # forward, as TimeAttnFn.forward does it
        wc, bc = wt @ wp, wt @ bp
        bias = bc + bt / c
        out = s_rows[:, None] * (o @ wc.t() + bias) + x                   # GEMM epilogue: scale * (acc + bias) + residual
        dropped = s_rows == 0
        out[dropped] = x[dropped] + bt                                    # vtx_dropped_rows_fix
        o_m = o.clone()
        o_m[dropped] = 0                                                  # ... which also zeroes the dropped rows of o
        assert torch.allclose(out, out_ref.detach(), rtol=1e-12, atol=1e-12)
        # backward, as TimeAttnFn.backward does it
        do = s_rows[:, None] * (dout @ wc)                                # scaled input-gradient GEMM
        G = dout.t() @ o_m                                                # ONE weight-gradient GEMM over the kept rows
        cs_all = dout.sum(0)
        u = c * (cs_all - dout[dropped].sum(0))                           # vtx_dropped_rows_colsum + ordered fold
        d_wt = c * (G @ wp.t()) + torch.outer(u, bp)
        d_bt = cs_all
        d_wp = c * (wt.t() @ G)
        d_bp = wt.t() @ u
    for got, p, name in ((d_wp, wp, 'proj.weight'), (d_bp, bp, 'proj.bias'), (d_wt, wt, 'temporal_fc.weight'),
                         (d_bt, bt, 'temporal_fc.bias'), (do, o_ref, 'attention output')):
        assert torch.allclose(got, p.grad, rtol=1e-10, atol=1e-10), name
    # the attention backward never needs the zeroed rows: their gradient is exactly zero
    assert do[dropped].abs().max() == 0
