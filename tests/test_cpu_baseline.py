"""The CPU baseline restatement computes the same step as the (golden-pinned) oracle."""
import torch

from oracle import clt_oracle as O
from oracle import cpu_baseline as B


def test_cpu_baseline_matches_oracle():
    L, d, M, k, N = 3, 16, 64, 4, 24
    layers = B.make_layers(L, d, M, cross=L)
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for lp in layers:
            for n, p in lp.items():
                if n.startswith("post_enc") or n.startswith("W_skip") or n.startswith("b_dec"):
                    p.copy_(0.1 * torch.randn(p.shape, generator=g))
    xs = [torch.randn(N, d, generator=g) for _ in range(L)]
    ys = [torch.randn(N, d, generator=g) for _ in range(L)]
    mask = torch.ones(N, dtype=torch.bool)
    mask[::5] = False
    ol = [O.LayerParams(lp["W_enc"].detach().clone(), lp["b_enc"].detach().clone(),
                        [lp[f"W_dec{t}"].detach().clone() for t in range(lp["T"])],
                        [lp[f"b_dec{t}"].detach().clone() for t in range(lp["T"])],
                        [lp[f"W_skip{t}"].detach().clone() for t in range(lp["T"])],
                        [lp[f"post_enc{t}"].detach().clone() for t in range(lp["T"])], None, lp["T"]) for lp in layers]
    res = O.clt_train_step(ol, xs, ys, mask, k)

    class NoOpt:
        def step(self): pass
        def zero_grad(self): pass

    fv = B.train_step(layers, xs, ys, mask, k, NoOpt())
    torch.testing.assert_close(torch.tensor(fv), torch.stack(res.fvus), rtol=1e-5, atol=1e-6)
    for l, lp in enumerate(layers):
        torch.testing.assert_close(lp["W_enc"].grad, res.grads[l]["W_enc"].float(), rtol=1e-4, atol=1e-6)
        torch.testing.assert_close(lp["b_enc"].grad, res.grads[l]["b_enc"].float(), rtol=1e-4, atol=1e-6)
        for t in range(lp["T"]):
            torch.testing.assert_close(lp[f"W_dec{t}"].grad, res.grads[l][f"W_decs.{t}"].float(), rtol=1e-4, atol=1e-6)
            torch.testing.assert_close(lp[f"post_enc{t}"].grad, res.grads[l][f"post_encs.{t}"].float(), rtol=1e-4, atol=1e-6)
            torch.testing.assert_close(lp[f"W_skip{t}"].grad, res.grads[l][f"W_skips.{t}"].float(), rtol=1e-4, atol=1e-6)
