"""The bound every GPU parity test of a CHAIN uses: the north star's 1e-4 plus a MEASURED allowance for
the fp32 reference's own rounding noise on the same input -- never a typed one.

``exact`` is the same quantity evaluated by the oracle in float64 from the same fp32 inputs (the
oracles are dtype-agnostic restatements).  max|ref32 - exact| measures how far the fp32 reference
itself is from exact arithmetic on this input: two fp32 evaluations with different summation orders
or fused multiply-adds cannot agree better than that, so k times it is added to the budget (k = 2 for
outputs, 4 for gradients, whose integrands jump at texel boundaries); on well-conditioned inputs it is
~1e-6 and the bound is the plain 1e-4.  The HIP result must also be no further from EXACT arithmetic
than the same bound.  Both distances are printed (pytest -s / on failure)."""
TOL = 1e-4


def close(a, b, tol=TOL, rel=False, what="", exact=None, outliers=0.0):
    """``outliers``: the fraction of ELEMENTS that may lie beyond the bound -- by at most 25 x -- in a comparison over
    randomly drawn cases (the seeded fuzz tests): a kink of the chain (|dist - prob| of the layout filter, the maximum over
    layers, a thresholded mask) that one input hits within an ulp flips a branch in one of three fp32 summation orders
    and moves the few elements downstream of it; a wrong kernel moves most of them."""
    if a is None or b is None:
        assert a is None and b is None, what
        return
    a = a.detach().cpu().double()
    b = b.detach().cpu().double()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    if a.numel() == 0:
        return
    scale = max(b.abs().max().item(), 1e-30) if rel else 1.0
    err = (a - b).abs().max().item()
    if exact is None:
        assert err <= tol * scale, f"{what}: max err {err:.3e} > {tol * scale:.3e}"
        return
    e64 = exact.detach().cpu().double()
    noise = (b - e64).abs().max().item()
    err64 = (a - e64).abs().max().item()
    bound = tol * scale + (4.0 if rel else 2.0) * noise
    print(f"[parity] {what}: |hip-ref32| {err:.3e}  |hip-ref64| {err64:.3e}  |ref32-ref64| {noise:.3e}  "
          f"(tol*scale {tol * scale:.1e})")
    if outliers > 0 and max(err, err64) > bound:
        beyond = (((a - b).abs() > bound) | ((a - e64).abs() > bound)).double().mean().item()
        print(f"[parity] {what}: {beyond:.2%} of the elements beyond the bound, the worst by {max(err, err64) / bound:.1f} x")
        assert beyond <= outliers and max(err, err64) <= 25 * bound, \
            f"{what}: {beyond:.2%} of the elements beyond {bound:.3e}, max err {max(err, err64):.3e}"
        return
    assert err <= bound, f"{what}: max err {err:.3e} > {bound:.3e} (tol*scale {tol * scale:.1e}, fp32 noise {noise:.1e})"
    assert err64 <= bound, (f"{what}: |hip - ref64| {err64:.3e} > {bound:.3e} "
                            f"(tol*scale {tol * scale:.1e}, |ref32 - ref64| {noise:.1e})")
