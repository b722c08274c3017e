"""vc_conv3d_fwd_noalloc (include/vc3d.h): the forward conv for callers that cannot allocate.  Its split-K partial slabs
live in a static device array inside libvc3d.so; everything else is vc_conv3d_fwd.  Called through the C ABI (ctypes)
and checked against vc_conv3d_fwd with a caller-owned workspace: the same kernels with the same split plan, so the
bf16 output must be IDENTICAL bit for bit (the bar for this path's deterministic split-K form); the BatchNorm sums
come from float atomics and agree to 1e-5 relative.  Two calls in a row, the second on a smaller input (the array
keeps the first call's contents), then an unsplit geometry, then one beyond the static size, which must be refused."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _case(N, Ci, T, H, W, Co, k, s, p, seed):
    from vc3d import capi, ops
    d = ops.conv_desc(N, Ci, T, H, W, Co, k, s, p)
    To, Ho, Wo = d.out_dims()
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = torch.randn(N, T, H, W, Ci, device="cuda", generator=g).to(torch.bfloat16)
    wf = (torch.randn(Co, k * k * k * Ci, device="cuda", generator=g) * (2.0 / (Co * k ** 3)) ** 0.5).to(torch.bfloat16)
    need = int(capi.lib.vc_conv3d_workspace_bytes(d, 0))
    # the existing entry point, with a caller-owned workspace of exactly the size it asks for
    ws = torch.empty(max(need, 16), dtype=torch.uint8, device="cuda")
    y_ref = torch.empty(N, To, Ho, Wo, Co, dtype=torch.bfloat16, device="cuda")
    st_ref = ops.new_stats(Co, x.device)
    capi.check(capi.lib.vc_conv3d_fwd(d, x.data_ptr(), wf.data_ptr(), y_ref.data_ptr(), st_ref.data_ptr(),
                                      ws.data_ptr() if need else None, need, capi.cur_stream()), "vc_conv3d_fwd")
    # the entry point that allocates nothing
    y = torch.full_like(y_ref, float("nan"))
    st = ops.new_stats(Co, x.device)
    rc = capi.lib.vc_conv3d_fwd_noalloc(d, x.data_ptr(), wf.data_ptr(), y.data_ptr(), st.data_ptr(), capi.cur_stream())
    torch.cuda.synchronize()
    return need, rc, y, y_ref, st, st_ref


def test_no_alloc_forward_equals_workspace_form_on_two_calls_then_refuses_oversize():
    from vc3d import capi
    limit = int(capi.lib.vc_static_scratch_bytes())
    assert limit == 64 << 20
    # first call: layer3-like, split-K (24.5 MiB of slabs); second: the same layer on a smaller batch (3 MiB)
    for N, seed in ((64, 1), (8, 2)):
        need, rc, y, y_ref, st, st_ref = _case(N, 256, 2, 7, 7, 256, 3, 1, 1, seed)
        assert 0 < need <= limit, need
        assert rc == 0, capi.lib.vc_last_error().decode()
        assert torch.equal(y.view(torch.int16), y_ref.view(torch.int16))
        assert torch.allclose(st.sum(0), st_ref.sum(0), rtol=1e-5, atol=1e-3)
    # a geometry that is not split: no scratch involved, same result
    need, rc, y, y_ref, st, st_ref = _case(4, 64, 8, 28, 28, 64, 3, 1, 1, 3)
    assert need == 0 and rc == 0
    assert torch.equal(y.view(torch.int16), y_ref.view(torch.int16))
    # beyond the static size: refused with a clear error, output untouched
    from vc3d import ops
    d = ops.conv_desc(128, 1024, 2, 7, 7, 2048, 3, 1, 1)
    assert int(capi.lib.vc_conv3d_workspace_bytes(d, 0)) > limit
    x = torch.zeros(128, 2, 7, 7, 1024, dtype=torch.bfloat16, device="cuda")
    w = torch.zeros(2048, 27 * 1024, dtype=torch.bfloat16, device="cuda")
    y = torch.full((128, 2, 7, 7, 2048), 7.0, dtype=torch.bfloat16, device="cuda")
    rc = capi.lib.vc_conv3d_fwd_noalloc(d, x.data_ptr(), w.data_ptr(), y.data_ptr(), None, capi.cur_stream())
    torch.cuda.synchronize()
    assert rc == -3 and "static scratch" in capi.lib.vc_last_error().decode()
    assert bool((y == 7.0).all())
