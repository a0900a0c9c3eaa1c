"""Unit tests of the decomposition's device helpers through the C ABI (die_pack.hip, die_ghost.hip): block pack /
unpack / max-merge, agent-record gather / scatter, and the ghost-refresh classification against numpy."""
import ctypes as C

import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def lib():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from die_amd import _lib
    return _lib


def _sp():
    from die_amd.device_array import stream_ptr
    return stream_ptr(torch.device('cuda:0'))


def test_rects_pack_unpack_and_max_merge(lib):
    rs = np.random.RandomState(1)
    a = torch.from_numpy(rs.rand(40, 24).astype(np.float32)).cuda()
    c = torch.from_numpy(rs.randint(0, 2 ** 62, (40, 24), dtype=np.int64)).cuda()
    blocks = [(a, 3, 9, 4, 20), (c, 0, 40, 0, 5), (a, 30, 40, 8, 24)]
    rects, off, want = [], 0, []
    for t, r0, r1, c0, c1 in blocks:
        rects.append(lib.Rect(t.data_ptr(), t.shape[1], r0, r1, c0, c1, t.element_size(), off))
        want.append((off, t[r0:r1, c0:c1].contiguous().view(torch.uint8).reshape(-1).cpu().numpy()))
        off += ((r1 - r0) * (c1 - c0) * t.element_size() + 7) & ~7
    arr = (lib.Rect * len(rects))(*rects)
    buf = torch.zeros(off, dtype=torch.uint8, device='cuda')
    lib.check(lib.lib.die_rects_pack(arr, len(rects), C.c_void_p(buf.data_ptr()), _sp()), 'pack')
    host = buf.cpu().numpy()
    for o, w in want:
        assert np.array_equal(host[o:o + len(w)], w)
    a0, c0_ = a.clone(), c.clone()
    a.zero_()
    c.zero_()
    lib.check(lib.lib.die_rects_unpack(arr, len(rects), C.c_void_p(buf.data_ptr()), _sp()), 'unpack')
    for t, ref, (_, r0, r1, cc0, cc1) in ((a, a0, blocks[0]), (c, c0_, blocks[1]), (a, a0, blocks[2])):
        assert torch.equal(t[r0:r1, cc0:cc1], ref[r0:r1, cc0:cc1])
    assert float(a[0, 0]) == 0.0                                   # cells outside the blocks untouched
    # max-merge on unsigned 64-bit words (a word with the top bit set must win over a small one)
    cur = torch.from_numpy(rs.randint(0, 2 ** 62, (40, 24), dtype=np.int64)).cuda()
    cur[0, 0] = -5                                                  # 0xFFFF…FB unsigned: larger than anything
    keep = cur.clone()
    r = (lib.Rect * 1)(lib.Rect(cur.data_ptr(), 24, 0, 40, 0, 5, 8, rects[1].buf_offset))
    lib.check(lib.lib.die_rects_unpack_max(r, 1, C.c_void_p(buf.data_ptr()), _sp()), 'unpack_max')
    got = cur[:, :5].cpu().numpy().view(np.uint64)
    assert np.array_equal(got, np.maximum(keep[:, :5].cpu().numpy().view(np.uint64), c0_[:, :5].cpu().numpy().view(np.uint64)))
    assert torch.equal(cur[:, 5:], keep[:, 5:])


def _arrays(n, rs):
    x = torch.from_numpy(rs.randint(-2 ** 31, 2 ** 31, n).astype(np.int32)).cuda()
    f = torch.from_numpy(rs.rand(n).astype(np.float32)).cuda()
    b = torch.from_numpy(rs.randint(0, 2, n).astype(np.uint8)).cuda()
    arrs = [x, f, b]
    ptrs = (C.c_void_p * 3)(*[t.data_ptr() for t in arrs])
    esz = (C.c_int32 * 3)(4, 4, 1)
    return arrs, ptrs, esz


def test_records_gather_scatter_roundtrip(lib):
    rs = np.random.RandomState(2)
    n = 5000
    arrs, ptrs, esz = _arrays(n, rs)
    idx = torch.from_numpy(rs.permutation(n)[:1234].astype(np.int64)).cuda()
    rec = torch.empty((3, 1234), dtype=torch.int32, device='cuda')
    lib.check(lib.lib.die_records_gather(ptrs, esz, 3, C.c_void_p(idx.data_ptr()), 1234, C.c_void_p(rec.data_ptr()), _sp()), 'gather')
    assert torch.equal(rec[0], arrs[0][idx]) and torch.equal(rec[1].view(torch.float32), arrs[1][idx])
    assert torch.equal(rec[2], arrs[2][idx].to(torch.int32))
    dst, dptr, _ = _arrays(n, np.random.RandomState(3))
    before = [t.clone() for t in dst]
    lib.check(lib.lib.die_records_scatter(dptr, esz, 3, C.c_void_p(idx.data_ptr()), 1234, C.c_void_p(rec.data_ptr()), _sp()), 'scatter')
    mask = torch.zeros(n, dtype=torch.bool, device='cuda')
    mask[idx] = True
    for t, src, old in zip(dst, arrs, before):
        assert torch.equal(t[mask], src[mask]) and torch.equal(t[~mask], old[~mask])


def test_records_gather_dev_and_scatter_at(lib):
    rs = np.random.RandomState(4)
    n, cap = 3000, 700
    arrs, ptrs, esz = _arrays(n, rs)
    for count in (0, 411, 700, 950):                               # below, at and above the message capacity
        idx = torch.from_numpy(rs.permutation(n)[:max(count, 1)].astype(np.int32)).cuda()
        cnt = torch.tensor([count], dtype=torch.int64, device='cuda')
        rec = torch.full((3, cap), -7, dtype=torch.int32, device='cuda')
        hdr = torch.zeros(1, dtype=torch.int64, device='cuda')
        lib.check(lib.lib.die_records_gather_dev(ptrs, esz, 3, C.c_void_p(idx.data_ptr()), C.c_void_p(cnt.data_ptr()), cap,
                                                 C.c_void_p(rec.data_ptr()), C.c_void_p(hdr.data_ptr()), _sp()), 'gather_dev')
        k = min(count, cap)
        assert int(hdr) == count
        sel = idx[:k].to(torch.int64)
        assert torch.equal(rec[0, :k], arrs[0][sel]) and torch.equal(rec[2, :k], arrs[2][sel].to(torch.int32))
        assert bool((rec[:, k:] == -7).all())
        if k:
            dst, dptr, _ = _arrays(n, np.random.RandomState(5))
            where = torch.from_numpy(np.random.RandomState(6).permutation(n)[:k].astype(np.int32)).cuda()
            lib.check(lib.lib.die_records_scatter_at(dptr, esz, 3, C.c_void_p(where.data_ptr()), k, cap, C.c_void_p(rec.data_ptr()), _sp()),
                      'scatter_at')
            w = where.to(torch.int64)
            assert torch.equal(dst[0][w], arrs[0][sel]) and torch.equal(dst[1][w], arrs[1][sel]) and torch.equal(dst[2][w], arrs[2][sel])


@pytest.mark.parametrize('grid,rank,halo', [((2, 2), 3, (12, 8)), ((1, 2), 0, (0, 16)), ((3, 1), 1, (10, 0)), ((2, 4), 5, (7, 4))])
def test_ghost_plan_matches_numpy_classification(lib, grid, rank, halo):
    from die_amd.device_array import to_q32
    from die_amd.dist import TileGeometry
    gW, gH = 48 * grid[0], 64 * grid[1]
    g = TileGeometry((gW, gH), grid, rank, halo)
    rs = np.random.RandomState(7)
    n = 20000
    qx, qy = to_q32(rs.rand(n)), to_q32(rs.rand(n))
    x = torch.from_numpy(qx.view(np.int32)).cuda()
    y = torch.from_numpy(qy.view(np.int32)).cuda()
    cx = ((qx.astype(np.uint64) * np.uint64(gW - 1) + np.uint64(2 ** 31)) >> np.uint64(32)).astype(np.int64)
    cy = ((qy.astype(np.uint64) * np.uint64(gH - 1) + np.uint64(2 ** 31)) >> np.uint64(32)).astype(np.int64)
    lx, ly = (cx - g.x0) % gW, (cy - g.y0) % gH
    keep = (lx < g.Wi) & (ly < g.Hi)
    near = {(-1, 0): lx < g.hx, (1, 0): lx >= g.Wi - g.hx, (0, -1): ly < g.hy, (0, 1): ly >= g.Hi - g.hy}
    want = []
    for dx, dy in g.DIRS:
        m = keep.copy()
        if dx:
            m &= near[(dx, 0)]
        if dy:
            m &= near[(0, dy)]
        want.append(np.nonzero(m)[0])
    want.append(np.nonzero(~keep)[0])
    nd = len(g.DIRS)
    plane = torch.zeros((g.W, g.H), dtype=torch.float32, device='cuda')
    own = torch.zeros((g.W, g.H), dtype=torch.int64, device='cuda')
    m = lib.Medium(g.W, g.H, lib.DIE_F32, 1, own.data_ptr(), plane.data_ptr(), plane.data_ptr(), plane.data_ptr(),
                   gW, gH, g.ox, g.oy, g.hx, g.hy, g.hx + g.Wi, g.hy + g.Hi, None)
    a = lib.Agents(n, x.data_ptr(), y.data_ptr(), None, None, None)
    caps = [n] * (nd + 1)
    caps[0] = 5                                                     # a list shorter than its population: truncated, total still exact
    lists = [torch.full((c,), -1, dtype=torch.int32, device='cuda') for c in caps]
    totals = torch.zeros(nd + 2, dtype=torch.int64, device='cuda')
    ws = torch.empty(lib.lib.die_ghost_workspace_bytes(n), dtype=torch.uint8, device='cuda')
    dirs = (C.c_int8 * (2 * nd))(*[v for d in g.DIRS for v in d])
    lib.check(lib.lib.die_ghost_plan(C.byref(m), C.byref(a), nd, dirs, (C.c_void_p * (nd + 1))(*[t.data_ptr() for t in lists]),
                                     (C.c_int64 * (nd + 1))(*caps), C.c_void_p(totals.data_ptr()), C.c_void_p(ws.data_ptr()), ws.numel(),
                                     _sp()), 'die_ghost_plan')
    t = totals.cpu().numpy()
    assert list(t[:nd + 1]) == [len(w) for w in want] and t[nd + 1] == keep.sum()
    for k, w in enumerate(want):
        got = lists[k].cpu().numpy()
        c = min(len(w), caps[k])
        assert np.array_equal(got[:c], w[:c]), f'list {k}'          # ascending array indices
        assert (got[c:] == -1).all()


@pytest.mark.parametrize('counts', [(40, 0, 25), (3, 2, 1), (0, 0, 0), (300, 300, 300)])
def test_ghost_pack_and_apply_against_numpy(lib, counts):
    """die_ghost_pack → (the buffer travels) → die_ghost_apply, with more / fewer / no arrivals than holes: arrivals
    overwrite the holes in order, surplus is appended, a cut tail is compacted into the remaining holes."""
    rs = np.random.RandomState(sum(counts) + 11)
    n, nd, caps = 2000, 3, [300, 300, 300]
    arrs, ptrs, esz = _arrays(n + 1000, rs)                       # capacity beyond n for appended arrivals
    F = 3
    # sender side: three lists of the given lengths
    lists = [torch.from_numpy(rs.permutation(n)[:max(c, 1)].astype(np.int32)).cuda() for c in counts]
    totals = torch.tensor(list(counts) + [0, 0], dtype=torch.int64, device='cuda')
    off, hdr, rec = 0, [], []
    for k in range(nd):
        hdr.append(off)
        rec.append(off + 16)
        off += 16 + F * caps[k] * 4
    buf = torch.zeros(off, dtype=torch.uint8, device='cuda')
    c64 = lambda v: (C.c_int64 * len(v))(*v)
    lp = (C.c_void_p * nd)(*[t.data_ptr() for t in lists])
    lib.check(lib.lib.die_ghost_pack(ptrs, esz, F, nd, lp, C.c_void_p(totals.data_ptr()), c64(caps), c64(hdr), c64(rec),
                                     C.c_void_p(buf.data_ptr()), _sp()), 'die_ghost_pack')
    host = buf.cpu().numpy()
    sent = []
    for k in range(nd):
        assert int(host[hdr[k]:hdr[k] + 8].view(np.int64)[0]) == counts[k]
        m = host[rec[k]:rec[k] + F * caps[k] * 4].view(np.int32).reshape(F, caps[k])[:, :counts[k]]
        sel = lists[k][:counts[k]].cpu().numpy().astype(np.int64)
        assert np.array_equal(m[0], arrs[0].cpu().numpy()[sel]) and np.array_equal(m[2], arrs[2].cpu().numpy()[sel].astype(np.int32))
        sent.append(m)
    # receiver side: its own arrays with H holes (ascending), a plan workspace whose mask marks them
    H = 90
    dst, dptr, _ = _arrays(n + 1000, np.random.RandomState(5))
    before = [t.cpu().numpy().copy() for t in dst]
    holes_np = np.sort(rs.permutation(n)[:H]).astype(np.int32)
    holes_np[-5:] = np.arange(n - 5, n)                            # some holes at the very end of the array
    holes_np = np.unique(holes_np)
    H = len(holes_np)
    holes = torch.from_numpy(holes_np).cuda()
    mask = np.full(n, 1 << 9, dtype=np.uint16)
    mask[holes_np] = 1 << 8
    ws = torch.from_numpy(mask.view(np.uint8)).cuda()
    rtot = torch.tensor([0, 0, 0, H, n - H], dtype=torch.int64, device='cuda')
    n_new = torch.zeros(3 + 2 * nd, dtype=torch.int64, device='cuda')
    lib.check(lib.lib.die_ghost_apply(dptr, esz, F, nd, C.c_void_p(rtot.data_ptr()), c64(caps), c64(hdr), c64(rec),
                                      C.c_void_p(buf.data_ptr()), C.c_void_p(holes.data_ptr()), C.c_void_p(ws.data_ptr()), n,
                                      n + 1000, C.c_void_p(n_new.data_ptr()), _sp()), 'die_ghost_apply')
    n_arr = sum(counts)
    arrivals = np.concatenate(sent, axis=1)                        # (F, n_arr) in side order
    want = [b.copy() for b in before]
    for f in range(F):
        view = want[f].view(np.int32) if want[f].dtype != np.uint8 else want[f]
        for j in range(n_arr):
            d = holes_np[j] if j < H else n + (j - H)
            view[d] = arrivals[f, j]
    if n_arr >= H:
        exp_n = n + n_arr - H
    else:
        exp_n = n - (H - n_arr)
        rest = holes_np[n_arr:]
        low = rest[rest < exp_n]
        tail = np.array([p for p in range(exp_n, n) if p not in set(holes_np.tolist())], dtype=np.int64)
        assert len(tail) == len(low)
        for f in range(F):
            want[f][low] = want[f][tail]
    assert n_new.cpu().tolist() == [exp_n, H, n - H, 0, 0, 0] + list(counts)
    for f in range(F):
        assert np.array_equal(dst[f].cpu().numpy()[:exp_n], want[f][:exp_n]), f'array {f}'


def test_ghost_apply_never_writes_past_the_capacity(lib):
    """More arrivals than holes + free capacity: the surplus is dropped, not written past the arrays, and the summary
    still carries the unclamped count so that the caller raises (DistEnv._refresh_ghosts_native)."""
    rs = np.random.RandomState(3)
    n, nd, caps, F = 500, 2, [400, 400], 3
    cap_arrays = n + 100                                           # 800 arrivals, 10 holes, room for 100
    src, sptr, esz = _arrays(1000, rs)
    lists = [torch.from_numpy(rs.permutation(1000)[:400].astype(np.int32)).cuda() for _ in range(nd)]
    totals = torch.tensor([400, 400, 0, 0], dtype=torch.int64, device='cuda')
    off, hdr, rec = 0, [], []
    for k in range(nd):
        hdr.append(off); rec.append(off + 16); off += 16 + F * caps[k] * 4
    buf = torch.zeros(off, dtype=torch.uint8, device='cuda')
    c64 = lambda v: (C.c_int64 * len(v))(*v)
    lp = (C.c_void_p * nd)(*[t.data_ptr() for t in lists])
    lib.check(lib.lib.die_ghost_pack(sptr, esz, F, nd, lp, C.c_void_p(totals.data_ptr()), c64(caps), c64(hdr), c64(rec),
                                     C.c_void_p(buf.data_ptr()), _sp()), 'die_ghost_pack')
    # receiver arrays of cap_arrays entries followed by a guard region that must stay untouched
    guard = 2000
    dst = [torch.full((cap_arrays + guard,), 0x5A5A5A5A, dtype=torch.int32, device='cuda'),
           torch.full((cap_arrays + guard,), 0x5A5A5A5A, dtype=torch.int32, device='cuda'),
           torch.full((cap_arrays + guard,), 0x5A, dtype=torch.uint8, device='cuda')]
    dptr = (C.c_void_p * F)(*[t.data_ptr() for t in dst])
    H = 10
    holes_np = np.arange(0, 10 * H, 10, dtype=np.int32)
    holes = torch.from_numpy(holes_np).cuda()
    mask = np.full(n, 1 << 9, dtype=np.uint16); mask[holes_np] = 1 << 8
    ws = torch.from_numpy(mask.view(np.uint8)).cuda()
    rtot = torch.tensor([0, 0, H, n - H], dtype=torch.int64, device='cuda')
    n_new = torch.zeros(3 + 2 * nd, dtype=torch.int64, device='cuda')
    lib.check(lib.lib.die_ghost_apply(dptr, esz, F, nd, C.c_void_p(rtot.data_ptr()), c64(caps), c64(hdr), c64(rec),
                                      C.c_void_p(buf.data_ptr()), C.c_void_p(holes.data_ptr()), C.c_void_p(ws.data_ptr()), n,
                                      cap_arrays, C.c_void_p(n_new.data_ptr()), _sp()), 'die_ghost_apply')
    torch.cuda.synchronize()
    assert int(n_new[0]) == n + 800 - H > cap_arrays               # the caller's `n_new > capacity` check fires
    for t in dst:
        g = t[cap_arrays:].cpu().numpy()
        assert (g == (0x5A if t.dtype == torch.uint8 else 0x5A5A5A5A)).all(), 'wrote past the capacity'
    # a capacity below the current count is refused up front
    rc = lib.lib.die_ghost_apply(dptr, esz, F, nd, C.c_void_p(rtot.data_ptr()), c64(caps), c64(hdr), c64(rec),
                                 C.c_void_p(buf.data_ptr()), C.c_void_p(holes.data_ptr()), C.c_void_p(ws.data_ptr()), n,
                                 n - 1, C.c_void_p(n_new.data_ptr()), _sp())
    assert rc == -1
