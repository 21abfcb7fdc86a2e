"""dev: in which order does the framework's reduction sum a short strided dimension (the L layers of alpha_ctx in
WIF._holes)?  Candidates against torch.sum on random data, counted in exactly equal outputs."""
import torch

dev = torch.device("cuda:0")
torch.manual_seed(0)
for shape, dim in (((1, 4, 10, 12, 128, 256), 3), ((1, 4, 10, 11, 128, 256), 3), ((2, 2, 3, 17, 64, 64), 3), ((1, 4, 10, 8, 128, 256), 3)):
    x = torch.rand(*shape, device=dev) * 2 - 1
    h = (x + 1) / 2
    want = h.sum(dim=dim)
    n = shape[dim]
    parts = [h.select(dim, i) for i in range(n)]

    def seq():
        acc = parts[0].clone()
        for p in parts[1:]:
            acc = acc + p
        return acc

    def accs(k, zero_start):
        a = [torch.zeros_like(parts[0]) if zero_start else None for _ in range(k)]
        for i, p in enumerate(parts):
            a[i % k] = p.clone() if a[i % k] is None else a[i % k] + p
        a = [v for v in a if v is not None]
        out = a[0]
        for v in a[1:]:
            out = out + v
        return out

    def accs_tree(k):
        a = [None] * k
        for i, p in enumerate(parts):
            a[i % k] = p.clone() if a[i % k] is None else a[i % k] + p
        a = [v for v in a if v is not None]
        while len(a) > 1:
            a = [a[i] + a[i + 1] if i + 1 < len(a) else a[i] for i in range(0, len(a), 2)]
        return a[0]

    cands = {"sequential": seq(), "4 accumulators, combined in order": accs(4, False), "2 accumulators": accs(2, False),
             "4 accumulators, tree": accs_tree(4), "8 accumulators, in order": accs(8, False), "8, tree": accs_tree(8)}
    print(shape, {k: f"{(v == want).double().mean().item():.6f}" for k, v in cands.items()})
    # the same on a strided view (the raw-slot layout: L planes inside a wider channel axis, Tc and Tp swapped)
    big = torch.rand(shape[0], shape[2], shape[1], shape[3] + 23, *shape[4:], device=dev) * 2 - 1
    view = big[:, :, :, 23:].permute(0, 2, 1, 3, 4, 5)
    hv = (view + 1) / 2
    wv = hv.sum(dim=dim)
    pv = [hv.select(dim, i) for i in range(n)]
    a = [None] * 4
    for i, p in enumerate(pv):
        a[i % 4] = p.clone() if a[i % 4] is None else a[i % 4] + p
    a = [v for v in a if v is not None]
    out = a[0]
    for v in a[1:]:
        out = out + v
    s = pv[0].clone()
    for p in pv[1:]:
        s = s + p
    print("   strided view:", f"4 acc {(out == wv).double().mean().item():.6f}", f"sequential {(s == wv).double().mean().item():.6f}")
