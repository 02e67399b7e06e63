"""CPU: the C-ABI library builds for gfx950, loads, exports every symbol include/splitvae.h
declares, and validates arguments without touching a GPU (no compute calls here)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from oracle import np_ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "splitvae.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sv_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree(lib_built):
    from split_vae_amd import _lib
    declared = header_functions()
    assert len(declared) >= 25
    assert sorted(_lib.SYMBOLS) == declared          # the ctypes table binds exactly the header's functions
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert b"gfx950" in lib.sv_version()


def test_library_is_gfx950_code_object(lib_built):
    data = open(lib_built, "rb").read()
    assert b"gfx950" in data and b"tile_conv_kernel" in data and b"wgrad_tile_kernel" in data


def test_param_table_matches_reference_order(lib_built):
    from split_vae_amd import _lib, ops
    for H in (32, 64):
        desc = _lib.LGVaeDesc(8, H, H, 128, 128, _lib.SV_BF16, 40.0)
        table = ops.param_table(desc)
        want = np_ref.param_shapes(H, H)
        assert [n for n, _, _ in table] == [n for n, _ in want]
        assert [tuple(s) for _, _, s in table] == [tuple(s) for _, s in want]
        offs = [o for _, o, _ in table]
        assert offs == sorted(offs) and all(o % 4 == 0 for o in offs)      # 16-byte aligned tensors
        n = _lib.load().sv_lgvae_param_count(C.byref(desc))
        assert n >= sum(int(np.prod(s)) for _, s in want) and n % 4 == 0


def test_plan_geometry_without_gpu(lib_built):
    from split_vae_amd import _lib
    lib = _lib.load()
    desc = _lib.LGVaeDesc(512, 64, 64, 128, 128, _lib.SV_BF16, 120.0)
    h = C.c_void_p()
    assert lib.sv_lgvae_plan_create(C.byref(desc), C.byref(h)) == 0
    ws = lib.sv_lgvae_workspace_bytes(h)
    assert 1 << 30 < ws < 8 << 30           # ~1.6 GB of activations for B=512: fits 288 GB many times over
    off, nb = C.c_int64(), C.c_int64()
    assert lib.sv_lgvae_buffer(h, b"out6_x", C.byref(off), C.byref(nb)) == 0
    assert nb.value == 512 * 64 * 64 * 6 * 4 and off.value % 256 == 0
    assert lib.sv_lgvae_buffer(h, b"no_such_buffer", C.byref(off), C.byref(nb)) == _lib.STATUS_BADARG
    # stepping an unbound plan is refused, not undefined behaviour
    a = _lib.StepArgs()
    a.phases = _lib.PHASE_ALL
    assert lib.sv_lgvae_step(h, C.byref(a), None) == _lib.STATUS_STATE
    lib.sv_lgvae_plan_destroy(h)


@pytest.mark.parametrize("gl,ll,B", [(128, 128, 512), (64, 64, 64), (256, 256, 96), (128, 128, 70), (256, 128, 64)])
def test_latent_slab_workspace_covers_every_split(lib_built, gl, ll, B):
    """the K-slice slabs of the latent block (lat_ws_x / lat_ws_xh) are sized from the four launches' own (split, N) pairs; a slab
    set is [split][B][N] fp32 with split >= 1, so each buffer holds at least one slab of the widest problem.  (The plan accepts equal
    power-of-two latent sizes only -- sv_lgvae_plan_create refuses the last case -- so the over-run the advisor describes cannot be reached.)"""
    from split_vae_amd import _lib
    lib = _lib.load()
    desc = _lib.LGVaeDesc(B, 64, 64, gl, ll, _lib.SV_BF16, 120.0)
    h = C.c_void_p()
    rc = lib.sv_lgvae_plan_create(C.byref(desc), C.byref(h))
    if gl != ll:
        assert rc == _lib.STATUS_UNSUPPORTED
        return
    assert rc == 0
    off, nb = C.c_int64(), C.c_int64()
    for name in (b"lat_ws_x", b"lat_ws_xh"):
        assert lib.sv_lgvae_buffer(h, name, C.byref(off), C.byref(nb)) == 0
        assert nb.value >= B * max(2 * gl, 2 * ll, gl + ll) * 4 and off.value % 256 == 0
    lib.sv_lgvae_plan_destroy(h)


@pytest.mark.parametrize("bad", [dict(H=48), dict(H=64, W=32), dict(gl=100), dict(dtype=7), dict(B=0)])
def test_plan_rejects_unsupported(lib_built, bad):
    from split_vae_amd import _lib
    lib = _lib.load()
    H = bad.get("H", 64)
    desc = _lib.LGVaeDesc(bad.get("B", 8), H, bad.get("W", H), bad.get("gl", 128), 128, bad.get("dtype", 1), 1.0)
    h = C.c_void_p()
    assert lib.sv_lgvae_plan_create(C.byref(desc), C.byref(h)) < 0
    assert lib.sv_lgvae_param_count(C.byref(desc)) < 0


def test_kernel_entry_points_validate_arguments(lib_built):
    from split_vae_amd import _lib
    lib = _lib.load()
    assert lib.sv_scramble_gather(None, None, None, 1, 32, 32, 4, None) == _lib.STATUS_BADARG
    assert lib.sv_random_perm(None, 1, 16, 0, 0, 0, None) == _lib.STATUS_BADARG
    assert lib.sv_adam_step(None, None, None, None, 16, 1e-4, .9, .999, 1e-7, 1, 1.0, None) == _lib.STATUS_BADARG
    d = _lib.ConvDesc(4, 24, 24, 32, 32, 4, 4, 1, 0, 1, 32, 32, 0)       # 24: not a power of two -> im2col kernels (SPLIT-SPAIR backbone)
    assert lib.sv_conv2d_wprep_elems(C.byref(d), 0) == 32 * 16 * 32
    d = _lib.ConvDesc(4, 50, 50, 32, 32, 4, 4, 3, 0, 1, 32, 32, 0)       # a stride that does not divide the extent
    assert lib.sv_conv2d_wprep_elems(C.byref(d), 0) < 0
    d = _lib.ConvDesc(4, 48, 48, 32, 32, 4, 4, 4, 0, 1, 32, 32, 0)       # strides 1..3
    assert lib.sv_conv2d_wprep_elems(C.byref(d), 0) < 0
    d = _lib.ConvDesc(4, 32, 32, 32, 6, 6, 6, 1, 0, 1, 32, 6, 1)
    # bf16 head (Cout 6): the x-pixel-packed image, 16 columns (2 pixels x 8) x 6x7 taps x Cin
    assert lib.sv_conv2d_wprep_elems(C.byref(d), 0) == 16 * 42 * 32
    d32 = _lib.ConvDesc(4, 32, 32, 32, 6, 6, 6, 1, 0, 0, 32, 6, 1)           # fp32 (the reference's precision): the same x-packed image since round 4
    assert lib.sv_conv2d_wprep_elems(C.byref(d32), 0) == 16 * 42 * 32
    assert lib.sv_conv2d_wprep_elems(C.byref(d), 1) == 32 * 36 * 8        # dgrad contracts over Cout padded to 8
    assert lib.sv_dlogistic_nll_workspace_bytes(8, 64, 64) == 8 * 4 * 4


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    """The product path must not fall back to anything when the HIP extension is absent."""
    from split_vae_amd import _lib
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    monkeypatch.setattr(_lib, "_lib", None)
    with pytest.raises(_lib.SplitVaeError):
        _lib.load()


def test_tape_host_logic_without_gpu(lib_built):
    """sv_tape_*: recording, argument checks, workspace layout -- pure host code (also what scripts/asan_host.sh runs under ASan + UBSan)."""
    from split_vae_amd import _lib
    lib = _lib.load()
    h = C.c_void_p()
    assert lib.sv_tape_create(C.byref(h), 0, _lib.SV_F32) == _lib.STATUS_BADARG
    assert lib.sv_tape_create(C.byref(h), 4, 7) == _lib.STATUS_BADARG
    assert lib.sv_tape_create(C.byref(h), 4, _lib.SV_F32) == 0
    x = lib.sv_tape_tensor(h, 64, 177, 180, 0)
    y = lib.sv_tape_tensor(h, 64, 64, 64, 1)
    z = lib.sv_tape_tensor(h, 64, 10, 12, 1)
    assert (x, y, z) == (0, 1, 2)
    assert lib.sv_tape_tensor(h, 64, 10, 8, 1) == _lib.STATUS_BADARG            # pitch below the width
    v = lib.sv_tape_view(h, y, 4, 1024, 1024)
    assert v == 3 and lib.sv_tape_view(h, y, 4, 2048, 2048) == _lib.STATUS_BADARG  # a view cannot outgrow its storage

    def node(kind, **kw):
        n = _lib.TapeNode()
        for f in ("x", "y", "t2", "t3", "t4", "t5", "t6"):
            setattr(n, f, -1)
        n.w_off = n.b_off = -1
        n.dyn_idx = n.loss_idx = -1
        n.rep, n.kind = 1, kind
        for k, val in kw.items():
            setattr(n, k, val)
        return lib.sv_tape_add(h, C.byref(n))

    assert node(_lib.TAPE_DENSE, x=x, y=y, w_off=0, b_off=177 * 64, act=_lib.SV_ACT_RELU) == 0
    assert node(_lib.TAPE_DENSE, x=x, y=y, w_off=-1) == _lib.STATUS_BADARG        # a Dense layer without a kernel
    assert node(_lib.TAPE_DENSE, x=y, y=z, w_off=12000, b_off=12640) == 0
    assert node(_lib.TAPE_UNARY, op=_lib.TAPE_SOFTPLUS, x=z, y=z, xo=1, yo=1, n=1) == 0
    assert node(_lib.TAPE_UNARY, op=_lib.TAPE_COPY, x=z, y=y, xo=0, yo=60, n=10) == _lib.STATUS_BADARG     # columns 60..69 of a 64-wide tensor
    assert node(_lib.TAPE_UNARY, op=99, x=z, y=z, n=1) == _lib.STATUS_BADARG
    assert node(_lib.TAPE_LOSS, loss_idx=0, mode=1, x=z, xo=0, t2=z, o2=1, R=16, n=1) == 0
    assert node(_lib.TAPE_LOSS, loss_idx=16, mode=1, x=z, t2=z, R=16, n=1) == _lib.STATUS_BADARG          # 16 loss slots
    assert node(_lib.TAPE_CONV, x=x, y=y, w_off=0, b_off=0, B=1, H=8, W=8, C=177, Cout=64, k=3, stride=1) != 0   # 177 channels: no conv geometry
    assert node(99) == _lib.STATUS_BADARG
    assert lib.sv_tape_workspace_bytes(h) == -1                                  # not finalized yet
    rep = (C.c_float * 16)(*([1.0] + [0.0] * 15))
    assert lib.sv_tape_set_report(h, rep, 1) == 0 and lib.sv_tape_set_report(h, rep, 17) == _lib.STATUS_BADARG
    assert lib.sv_tape_finalize(h) == 0 and lib.sv_tape_finalize(h) == _lib.STATUS_BADARG
    ws = lib.sv_tape_workspace_bytes(h)
    assert ws > 64 * (180 + 64 + 12) * 4 and ws % 256 == 0
    off, goff = C.c_int64(), C.c_int64()
    assert lib.sv_tape_tensor_info(h, x, C.byref(off), C.byref(goff)) == 0 and goff.value == -1 and off.value % 256 == 0
    assert lib.sv_tape_tensor_info(h, v, C.byref(off), C.byref(goff)) == 0 and goff.value >= 0
    assert lib.sv_tape_tensor_info(h, 9, None, None) == _lib.STATUS_BADARG
    assert lib.sv_tape_bind(h, None, ws, None) == _lib.STATUS_BADARG
    assert lib.sv_tape_run(h, None, None) == _lib.STATUS_BADARG
    lib.sv_tape_destroy(h)


def test_tape_with_conv_layers_reserves_the_weight_gradient_slabs(lib_built):
    """A tape that holds a Conv2D node reserves sv_conv2d_wgrad_workspace_bytes for the LDS-tile weight gradients' partial-sum slabs (one region,
    shared by its layers in stream order); a tape of Dense layers only does not.  Host logic: no GPU."""
    from split_vae_amd import _lib
    lib = _lib.load()

    def build(with_conv):
        h = C.c_void_p()
        assert lib.sv_tape_create(C.byref(h), 4, _lib.SV_F32) == 0
        x = lib.sv_tape_tensor(h, 4 * 16 * 16, 32, 32, 0)
        y = lib.sv_tape_tensor(h, 4 * 16 * 16, 32, 32, 1)
        n = _lib.TapeNode()
        for f in ("x", "y", "t2", "t3", "t4", "t5", "t6"):
            setattr(n, f, -1)
        n.dyn_idx = n.loss_idx = -1
        n.rep = 1
        n.x, n.y, n.w_off = x, y, 0
        if with_conv:
            n.kind, n.b_off = _lib.TAPE_CONV, 3 * 3 * 32 * 32
            n.B, n.H, n.W, n.C, n.Cout, n.k, n.stride = 4, 16, 16, 32, 32, 3, 1
        else:
            n.kind, n.b_off = _lib.TAPE_DENSE, 32 * 32
        assert lib.sv_tape_add(h, C.byref(n)) == 0
        assert lib.sv_tape_finalize(h) == 0
        ws = lib.sv_tape_workspace_bytes(h)
        lib.sv_tape_destroy(h)
        return ws

    d = _lib.ConvDesc()
    d.B, d.H, d.W, d.Cin, d.Cout, d.KH, d.KW, d.stride = 4, 16, 16, 32, 32, 3, 3, 1
    d.dtype, d.ldx, d.ldy = _lib.SV_F32, 32, 32
    slabs = lib.sv_conv2d_wgrad_workspace_bytes(C.byref(d))
    assert slabs > 0
    assert build(True) >= slabs and build(False) < slabs


def test_tape_lane_schedule_without_gpu(lib_built):
    """The cross-lane dependencies sv_tape_finalize derives from the nodes' tensors (csrc/tape.hip: build_schedules), on a diamond: x -> A (lane 0) -> ya,
    x -> B (lane 1) -> yb, C (lane 0) = concat(ya, yb) -> z, loss on z.  Forward: C waits for B (read-after-write across lanes), nothing else crosses.  Backward
    (reverse order: loss, C, B, A): B (lane 1) waits for C's adjoint -- it wrote grad(yb) --, and A (lane 0) waits for B: both ADD into grad(x), and conflicting
    accumulations keep the tape's order.  Pure host code: no GPU."""
    import ctypes as C
    from split_vae_amd import _lib
    lib = _lib.load()
    if os.environ.get("SV_TAPE_LANES") == "0":
        pytest.skip("lanes are switched off in this environment")
    h = C.c_void_p()
    assert lib.sv_tape_create(C.byref(h), 4, _lib.SV_F32) == 0
    x = lib.sv_tape_tensor(h, 16, 32, 32, 1)
    ya = lib.sv_tape_tensor(h, 16, 8, 8, 1)
    yb = lib.sv_tape_tensor(h, 16, 8, 8, 1)
    z = lib.sv_tape_tensor(h, 16, 16, 16, 1)

    def node(kind, **kw):
        n = _lib.TapeNode()
        for f in ("x", "y", "t2", "t3", "t4", "t5", "t6"):
            setattr(n, f, -1)
        n.w_off = n.b_off = -1
        n.dyn_idx = n.loss_idx = -1
        n.rep, n.kind = 1, kind
        for k, val in kw.items():
            setattr(n, k, val)
        assert lib.sv_tape_add(h, C.byref(n)) == 0

    node(_lib.TAPE_DENSE, x=x, y=ya, w_off=0, b_off=256, lane=0)                      # node 0: A
    node(_lib.TAPE_DENSE, x=x, y=yb, w_off=512, b_off=768, lane=1)                    # node 1: B
    node(_lib.TAPE_UNARY, op=_lib.TAPE_COPY, x=ya, y=z, xo=0, yo=0, n=8, group=1)     # nodes 2, 3: C = one launch (a group)
    node(_lib.TAPE_UNARY, op=_lib.TAPE_COPY, x=yb, y=z, xo=0, yo=8, n=8, group=1)
    node(_lib.TAPE_LOSS, loss_idx=0, mode=1, x=z, xo=0, t2=z, o2=8, R=4, n=8)         # node 4
    bad = _lib.TapeNode()
    bad.kind, bad.x, bad.y, bad.lane = _lib.TAPE_UNARY, z, z, 9
    assert lib.sv_tape_add(h, C.byref(bad)) == _lib.STATUS_BADARG                     # lanes 0 .. 3
    assert lib.sv_tape_schedule(h, 0, 0, None, 0, None) == _lib.STATUS_BADARG         # not finalized
    assert lib.sv_tape_finalize(h) == 0

    def sched(p, nd):
        w = (C.c_int32 * 8)()
        rec = C.c_int32()
        k = lib.sv_tape_schedule(h, p, nd, w, 8, C.byref(rec))
        assert k >= 0
        return sorted(w[i] for i in range(k)), rec.value

    fwd = [sched(0, i) for i in range(5)]
    bwd = [sched(1, i) for i in range(5)]
    # forward: B waits for nothing (x is an input); the group C (waits on its first node, 2) waits for B's event; B records, nobody else
    assert fwd[0] == ([], 0) and fwd[1] == ([], 1) and fwd[2] == ([1], 0) and fwd[3][0] == [] and fwd[4] == ([], 0), fwd
    # backward: the group C's waits sit on its LAST node (3): none; its event behind its FIRST node (2), which B waits for; A waits for B
    assert bwd[4] == ([], 0) and bwd[3][0] == [] and bwd[2] == ([], 1) and bwd[1] == ([2], 1) and bwd[0] == ([1], 0), bwd
    assert lib.sv_tape_schedule(h, 2, 0, None, 0, None) == _lib.STATUS_BADARG and lib.sv_tape_schedule(h, 0, 5, None, 0, None) == _lib.STATUS_BADARG
    lib.sv_tape_destroy(h)
