"""csrc/records.hip (pack / unpack of the visible-splat records of the Gaussian-sharded renderer) against the torch formulation
it replaces (`distributed.pack_visible` / `unpack_records`, themselves the reference's concat + mask + split,
internal/renderers/gsplat_distributed_renderer.py:313-414): records bit-identical, gradients identical up to summation order."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _inputs(C, N, seed, batched):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)
    radii = (torch.rand(C, N, generator=g) * 9).to(torch.int32)
    radii[radii < 3] = 0                                     # a third of the pairs invisible
    radii[:, N - 1] = 0 if seed % 2 else 5                   # the last splat of a camera both ways
    base = dict(means2d=r(C, N, 2), depths=r(C, N).abs() + 0.1, conics=r(C, N, 3), comps=torch.rand(C, N, generator=g), rgbs=r(C, N, 3))
    opac = torch.rand(N, 1, generator=g)
    radii = radii.to(DEV)
    base = {k: v.to(DEV).requires_grad_(True) for k, v in base.items()}
    opac = opac.to(DEV).requires_grad_(True)
    if batched:
        per_cam = {k: [v[c] for c in range(C)] for k, v in base.items()}
    else:                                                    # separate tensors per camera (the non-batched projection)
        per_cam = {k: [v[c].clone() for c in range(C)] for k, v in base.items()}
    results = [(radii[c], per_cam["means2d"][c], per_cam["depths"][c], per_cam["conics"][c], per_cam["comps"][c], radii[c] > 0) for c in range(C)]
    return results, per_cam["rgbs"], opac, base


@pytest.mark.parametrize("C,N", [(1, 5000), (3, 4097), (2, 1)])
@pytest.mark.parametrize("batched", [True, False])
@pytest.mark.parametrize("fold", [True, False])
def test_pack_unpack_match_the_torch_formulation(C, N, batched, fold):
    import gspl_amd  # noqa: F401
    from gspl_amd import distributed as D, ops

    def run(hip):
        results, rgbs, opac, base = _inputs(C, N, seed=C * 7 + N, batched=batched)
        leaves = [opac] + [t for r in results for t in r[1:5]] + list(rgbs)
        for t in leaves:
            if not t.is_leaf:
                t.retain_grad()
        if hip:
            records, counts = ops.pack_visible_records(results, rgbs, opac)
            radii, means2d, depths, conics, op, col = ops.unpack_visible_records(records, fold)
        else:
            recs = [D.pack_visible(r[0], r[1], r[2], r[3], r[4], opac, rgb, r[5]) for r, rgb in zip(results, rgbs)]
            counts = [int(x.shape[0]) for x in recs]
            records = torch.cat(recs, dim=0)
            radii, means2d, depths, conics, comps, op, col = D.unpack_records(records)
            op = (op * comps.unsqueeze(-1) if fold else op).squeeze(-1)
        g = torch.Generator().manual_seed(5)
        w = [torch.randn(t.shape, generator=g).to(DEV) for t in (means2d, depths, conics, op, col)]
        loss = sum((a * b).sum() for a, b in zip((means2d, depths, conics, op, col), w))
        loss.backward()
        grads = [t.grad for t in leaves]
        return records.detach(), counts, radii, [t.detach() for t in (means2d, depths, conics, op, col)], grads

    rec_h, cnt_h, rad_h, out_h, g_h = run(True)
    rec_t, cnt_t, rad_t, out_t, g_t = run(False)
    assert cnt_h == cnt_t and sum(cnt_h) == rec_h.shape[0]
    assert torch.equal(rec_h.view(torch.int32), rec_t.view(torch.int32))
    assert torch.equal(rad_h, rad_t)
    for a, b in zip(out_h, out_t):
        assert torch.equal(a, b)
    for i, (a, b) in enumerate(zip(g_h, g_t)):
        if b is None:
            assert a is None or float(a.abs().max()) == 0.0
            continue
        assert a is not None, i
        assert torch.allclose(a, b, rtol=1e-6, atol=1e-6), (i, float((a - b).abs().max()))


def test_no_visible_splat_at_all():
    import gspl_amd  # noqa: F401
    from gspl_amd import ops
    C, N = 2, 300
    z = lambda *s: torch.zeros(*s, device=DEV)
    results = [(torch.zeros(N, dtype=torch.int32, device=DEV), z(N, 2), z(N), z(N, 3), z(N), None) for _ in range(C)]
    records, counts = ops.pack_visible_records(results, [z(N, 3) for _ in range(C)], z(N, 1))
    assert counts == [0, 0] and records.shape == (0, 12)
    radii, means2d, *_ = ops.unpack_visible_records(records, True)
    assert radii.shape == (0,) and means2d.shape == (0, 2)


@pytest.mark.parametrize("C,N", [(1, 5000), (3, 4097), (2, 1)])
def test_two_phase_pack_is_the_one_phase_pack(C, N):
    """`gspl_records_count_fwd` + `gspl_records_scatter_fwd` (counts on their way to the host before the colours exist: the
    three-node sharded step, ops.sharded_front) leave the same slots, ends and records as `gspl_records_pack_fwd`."""
    import gspl_amd  # noqa: F401
    from gspl_amd import _lib as L, ops
    results, rgbs, opac, base = _inputs(C, N, seed=C * 11 + N, batched=True)
    records_ref, counts_ref = ops.pack_visible_records(results, rgbs, opac)
    radii = torch.stack([r[0] for r in results]).contiguous()
    t = {k: v.detach().contiguous() for k, v in base.items()}
    op = opac.detach().reshape(-1).contiguous()
    slots = torch.empty((C, N), dtype=torch.int32, device=DEV)
    ends = torch.empty((C,), dtype=torch.int64, device=DEV)
    host_ends = torch.empty((C,), dtype=torch.int64).pin_memory()
    ws_bytes = L.lib().gspl_records_workspace_bytes(C, N)
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=DEV)
    records = torch.full((C * N, L.GSPL_RECORD_FLOATS), float("nan"), device=DEV)
    with torch.cuda.device(0):
        L.call("gspl_records_count_fwd", C, N, L.ptr(radii), L.ptr(slots), L.ptr(ends), host_ends.data_ptr(), L.ptr(ws), ws_bytes, L.stream())
        torch.cuda.synchronize()
        host = [int(v) for v in host_ends.tolist()]                 # known BEFORE the scatter is launched
        L.call("gspl_records_scatter_fwd", C, N, L.ptr(radii), L.ptr(slots), L.ptr(t["means2d"]), L.ptr(t["depths"]), L.ptr(t["conics"]),
               L.ptr(t["comps"]), L.ptr(op), L.ptr(t["rgbs"]), L.ptr(records), L.stream())
    assert host == [int(v) for v in ends.tolist()]
    e = [0] + host
    assert [e[i + 1] - e[i] for i in range(C)] == counts_ref
    assert torch.equal(records[:host[-1]].view(torch.int32), records_ref.detach().view(torch.int32))
    assert bool(torch.isnan(records[host[-1]:]).all())               # nothing written past the last record
    vis = radii > 0
    assert torch.equal(slots >= 0, vis) and torch.equal(slots[vis].long(), torch.arange(host[-1], device=DEV))


@pytest.mark.parametrize("C,N", [(1, 5000), (3, 4097), (2, 1)])
def test_pad_kernel_is_the_torch_formulation_of_the_fixed_size_format(C, N):
    """`gspl_records_pad_fwd` (three-node sharded step, padded exchange) writes the rows `ops.pack_all_records` builds with
    concat + where, bit for bit, and the identity slots its backward takes."""
    import gspl_amd  # noqa: F401
    from gspl_amd import _lib as L, ops
    results, rgbs, opac, base = _inputs(C, N, seed=C * 13 + N, batched=True)
    ref = ops.pack_all_records(results, rgbs, opac).detach()
    radii = torch.stack([r[0] for r in results]).contiguous()
    t = {k: v.detach().contiguous() for k, v in base.items()}
    op = opac.detach().reshape(-1).contiguous()
    records = torch.full((C * N, L.GSPL_RECORD_FLOATS), float("nan"), device=DEV)
    slots = torch.empty((C, N), dtype=torch.int32, device=DEV)
    with torch.cuda.device(0):
        L.call("gspl_records_pad_fwd", C, N, L.ptr(radii), L.ptr(t["means2d"]), L.ptr(t["depths"]), L.ptr(t["conics"]), L.ptr(t["comps"]),
               L.ptr(op), L.ptr(t["rgbs"]), L.ptr(records), L.ptr(slots), L.stream())
    assert torch.equal(records.view(torch.int32), ref.view(torch.int32))
    ident = torch.arange(C * N, dtype=torch.int32, device=DEV).reshape(C, N)
    assert torch.equal(slots, torch.where(radii > 0, ident, -1))
