"""Error metrics shared by the GPU parity tests.

`relerr` (max |a-b| / max |b|) alone lets a tensor pass whose large entries are right and whose small entries are garbage.  `assert_close`
therefore checks three things against the same tolerance `rtol`:
  1. max-norm:   max|a-b| <= rtol * max|b|                       (the old metric)
  2. relative L2: ||a-b||_2 <= l2_tol * ||b||_2                   (l2_tol defaults to rtol)
  3. elementwise: |a-b| <= rtol * |b| + rtol * rms(b)  everywhere (torch.allclose with atol = rtol * rms(ref)): an entry far below the
     RMS of the tensor still has to be right to rtol * rms, not to rtol * max.
north_star: "within 1e-3 relative" (fp32); the bf16 tolerances are the bf16-rounding bounds stated in each test."""
import os

import torch


def relerr(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


def rel_l2(a, b):
    a, b = a.detach().double().cpu().flatten(), b.detach().double().cpu().flatten()
    return ((a - b).norm() / (b.norm() + 1e-300)).item()


def worst_elementwise(a, b, rtol):
    """max over elements of |a-b| / (rtol*|b| + rtol*rms(b)): <= 1 iff torch.allclose(a, b, rtol, atol=rtol*rms(b))"""
    a, b = a.detach().double().cpu().flatten(), b.detach().double().cpu().flatten()
    rms = b.pow(2).mean().sqrt().item()
    if rms == 0.0:
        return 0.0 if (a - b).abs().max().item() == 0.0 else float("inf")
    return ((a - b).abs() / (rtol * b.abs() + rtol * rms)).max().item()


def assert_close(a, b, rtol, name="", l2_tol=None, elem_mult=1.0):
    """elem_mult > 1 loosens ONLY check 3 (used where a tensor has legitimately sign-sensitive elements, with the reason at the call)"""
    assert a.shape == b.shape, f"{name}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    e_max, e_l2 = relerr(a, b), rel_l2(a, b)
    e_el = worst_elementwise(a, b, rtol * elem_mult)
    if os.environ.get("NMH_PRINT_ERR"):
        print(f"  [{name}] max-norm {e_max:.2e}  rel-L2 {e_l2:.2e}  worst elementwise / tol {e_el:.2f}  (rtol {rtol:g})")
    assert e_max < rtol, f"{name}: max-norm rel err {e_max:.3e} >= {rtol:g}"
    assert e_l2 < (rtol if l2_tol is None else l2_tol), f"{name}: relative L2 err {e_l2:.3e} >= {rtol if l2_tol is None else l2_tol:g}"
    assert e_el <= 1.0, f"{name}: elementwise |a-b| exceeds rtol*|b| + rtol*rms(b) by {e_el:.2f}x (rtol {rtol * elem_mult:g})"
