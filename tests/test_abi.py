"""The C-ABI boundary: libase_hip.so loads on a machine without a GPU and exports every function that
include/ase_hip.h declares, with the arity ase_amd/lib.py binds (no compute calls here)."""
import ctypes
import os
import re

import pytest

from ase_amd import lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_decls():
    src = open(os.path.join(ROOT, 'include', 'ase_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    decls = {}
    for m in re.finditer(r'\b(?:int|const char\*)\s+(ase_hip_\w+)\s*\(([^;]*?)\)\s*;', src, flags=re.S):
        args = m.group(2).strip()
        n = 0 if args in ('void', '') else len([a for a in args.split(',') if a.strip()])
        decls[m.group(1)] = n
    return decls


def test_library_is_built_and_loads():
    assert os.path.exists(L.LIB_PATH), "run __graft_entry__.build() first"
    lib = L.load()
    assert lib.ase_hip_abi_version() == L.ABI_VERSION
    assert lib.ase_hip_last_error() is not None


def test_every_declared_symbol_is_exported_and_bound():
    decls = _header_decls()
    assert len(decls) >= 25
    lib = ctypes.CDLL(L.LIB_PATH)
    for name, nargs in decls.items():
        assert hasattr(lib, name), f'{name} declared in ase_hip.h but not exported'
        if name in ('ase_hip_abi_version', 'ase_hip_last_error'):
            continue
        assert name in L.SIGNATURES, f'{name} has no ctypes binding in ase_amd/lib.py'
        assert len(L.SIGNATURES[name]) == nargs, (name, len(L.SIGNATURES[name]), nargs)
    for name in L.SIGNATURES:
        assert name in decls, f'{name} bound in lib.py but not declared in the header'


def test_argument_validation_fails_loudly_without_gpu():
    """Null operands are rejected by the host-side checks before any launch (works without a GPU)."""
    lib = L.load()
    rc = lib.ase_hip_gemm_nt(None, 0, None, 0, None, 0, None, None, 0, 0, 0, None, 0, None, 0, 0, 0, 0, 0, 0, 0, 1.0, None, L.BF16, None)
    assert rc == -1 and b'gemm_nt' in lib.ase_hip_last_error()
    with pytest.raises(L.AseHipError):
        L.check(rc, 'gemm_nt')


def test_product_has_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from ase_amd.backend import HipBackend
    with pytest.raises(L.AseHipError):
        HipBackend()


def _plan(problems, target=256):
    """problems: [(M, N, K, n_real, k_real, bias_rows)] -> (rc, work items) through the host-only planner."""
    lib = L.load()
    n = len(problems)
    tab = (ctypes.c_int64 * (16 * n))()
    for i, (M, N, K, nr, kr, br) in enumerate(problems):
        row = [0x1000, N, 0x2000, K, 0x3000, 0, br, M, N, K, nr, kr, kr, kr, 0x3F800000, 0]
        for j, v in enumerate(row):
            tab[16 * i + j] = v
    work = (ctypes.c_int32 * (4 * 8192))()
    nw = ctypes.c_int(0)
    red = (ctypes.c_int32 * (4 * 8192))()
    nr = ctypes.c_int(0)
    rc = lib.ase_hip_gemm_tn_grouped_plan(tab, n, target, work, 8192, ctypes.byref(nw), red, 8192, ctypes.byref(nr))
    raw = [tuple(work[4 * i:4 * i + 4]) for i in range(nw.value)]
    # launch order, padded per XCD range with empty items (K-tiles = 0); field 3 = K-tiles | workspace slab << 16
    items = [(p_, t_, m0, w3 & 0xFFFF) for (p_, t_, m0, w3) in raw if w3 & 0xFFFF]
    slabs = [w3 >> 16 for (p_, t_, m0, w3) in raw if w3 & 0xFFFF]
    if rc == 0:
        assert nw.value % 8 == 0 and sorted(slabs) == list(range(len(slabs)))          # every slab exactly once
        # reduce list: one entry per (problem, tile); split s of the tile wrote slab first + s * tiles(problem)
        ents = [tuple(red[4 * i:4 * i + 4]) for i in range(nr.value)]
        tiles_of = {}
        for (p_, t_, m0, nk) in items:
            tiles_of[p_] = max(tiles_of.get(p_, 0), t_ + 1)
        assert sorted((p_, t_) for p_, t_, _, _ in ents) == sorted({(p_, t_) for p_, t_, _, _ in items})
        for p_, t_, first, splits in ents:
            own = sorted(sl for sl, it in zip(slabs, items) if it[0] == p_ and it[1] == t_)
            assert own == [first + s_ * tiles_of[p_] for s_ in range(splits)], (p_, t_, first, splits, own)
        # whole sharing groups per XCD range where the packing allows it: the tiles of one (problem, row range) of a power-of-
        # two tile count never straddle two ranges when everything fits one round
        cap = nw.value // 8
        where = {}
        for pos, (p_, t_, m0, w3) in enumerate(raw):
            if w3 & 0xFFFF:
                where.setdefault((p_, m0), set()).add(pos // cap)
        _plan.straddlers = sum(len(v) > 1 for v in where.values())
    return rc, items, tab


def test_grouped_weight_gradient_plan_covers_every_tile_once():
    """Host-side planner of ase_hip_gemm_tn_grouped (no GPU): every 256 x 256 tile of every layer gets disjoint row
    ranges that add up to M, in whole 64-row K-tiles, and the config-2 layer set lands on one workgroup per CU."""
    probs = [(32768, 1024, 320, 1024, 317, 0), (32768, 1024, 1024, 1024, 1024, 0), (32768, 512, 1024, 512, 1024, 0),
             (16384, 1024, 320, 1024, 317, 0), (16384, 1024, 1024, 1024, 1024, 0), (16384, 512, 1024, 512, 1024, 0),
             (16384, 1024, 1408, 1024, 1400, 12288), (16384, 1024, 1024, 1024, 1024, 12288), (16384, 512, 1024, 512, 1024, 12288)]
    rc, items, tab = _plan(probs)
    assert rc == 0
    assert len(items) == 256 and _plan.straddlers == 0       # one workgroup per CU, no operand panel fetched by two XCDs
    cover = {}
    for p, t, m0, nk in items:
        M, N, K, nr, kr, br = probs[p]
        tiles = ((nr + 255) // 256) * ((K + 255) // 256)
        assert 0 <= t < tiles and m0 % 64 == 0 and nk >= 1 and m0 + 64 * nk <= M
        cover.setdefault((p, t), []).append((m0, m0 + 64 * nk))
    for p, (M, N, K, nr, kr, br) in enumerate(probs):
        tiles = ((nr + 255) // 256) * ((K + 255) // 256)
        assert (tab[16 * p + 15] & 0xFFFF) == (K + 255) // 256 and tab[16 * p + 6] == (br or M)
        for t in range(tiles):
            segs = sorted(cover[(p, t)])
            assert segs[0][0] == 0 and segs[-1][1] == M
            assert all(a[1] == b[0] for a, b in zip(segs, segs[1:]))
    # fewer tiles than workgroups per layer set, longer contraction: still a full cover with no split below 4 K-tiles
    rc, items, _ = _plan([(1024, 256, 256, 200, 250, 0)])
    assert rc == 0 and sum(nk for _, _, _, nk in items) == 16 and all(nk >= 4 for *_, nk in items)


def test_grouped_plan_rejects_ragged_rows():
    rc, _, _ = _plan([(1000, 256, 256, 256, 256, 0)])
    assert rc == -1 and b'64-row' in L.load().ase_hip_last_error()
    rc, _, _ = _plan([(1024, 256, 256, 256, 256, 100)])
    assert rc == -1


def test_nt_kernel_choice_is_reported():
    lib = L.load()
    assert lib.ase_hip_gemm_nt_kernel_id(16384, 1024, 1024, L.BF16) == 2      # phased 256 x 256
    assert lib.ase_hip_gemm_nt_kernel_id(16384, 1024, 1024, L.F32) == 3       # lock-step 256 x 256 (exact f32)
    assert lib.ase_hip_gemm_nt_kernel_id(16384, 512, 1024, L.BF16) == 1       # 128 x 128 (512 workgroups)
    assert lib.ase_hip_gemm_nt_kernel_id(4096, 1024, 1024, L.BF16) == 4       # 64 x 128: 128 x 128 would give 256 workgroups
    assert lib.ase_hip_gemm_nt_kernel_id(4096, 512, 1024, L.BF16) == 5        # 64 x 64
    assert lib.ase_hip_gemm_nt_kernel_id(2048, 1024, 1024, L.BF16) == 5       # one rank's shard at 8 GPUs
    assert lib.ase_hip_gemm_nt_kernel_id(16384, 64, 512, L.BF16) == 0         # narrow head
    # single rounds of 192-255 tiles of 256 x 256 that become <= 256 tiles of 192 x 256: the phased kernel's 192-row variant
    assert lib.ase_hip_gemm_nt_kernel_id(12288, 1024, 1408, L.F16) == 6        # the discriminator's 3 x 4096 rows: 192 -> 256 workgroups
    assert lib.ase_hip_gemm_nt_kernel_id(6144, 2048, 512, L.BF16) == 6
    assert lib.ase_hip_gemm_nt_kernel_id(12288, 1024, 1408, L.F32) == 3        # (4-byte storage: lock-step)
    assert lib.ase_hip_gemm_nt_kernel_id(12288, 512, 1024, L.F16) == 4         # 96 tiles: not a phased grid


def test_grouped_plan_random_layer_sets():
    """Property check of the host-side planner over random layer sets: disjoint 64-row-aligned ranges that tile [0, M) for
    every output tile, never more work items than the table holds, deterministic."""
    import random
    rnd = random.Random(5)
    for trial in range(40):
        probs = []
        for _ in range(rnd.randint(1, 12)):
            M = 64 * rnd.randint(1, 600)
            nr, kr = rnd.choice([64, 128, 200, 512, 1000, 1024]), rnd.choice([128, 317, 512, 1024, 1400])
            N, K = (nr + 63) // 64 * 64, (kr + 63) // 64 * 64
            probs.append((M, N, K, nr, kr, rnd.choice([0, 0, 64 * rnd.randint(1, M // 64)])))
        target = rnd.choice([64, 256, 512])
        rc, items, _ = _plan(probs, target)
        rc2, items2, _ = _plan(probs, target)
        assert rc == 0 and items == items2
        cover = {}
        for p, t, m0, nk in items:
            M, N, K, nr, kr, br = probs[p]
            assert 0 <= t < ((nr + 255) // 256) * ((K + 255) // 256)
            assert m0 % 64 == 0 and nk >= 1 and m0 + 64 * nk <= M
            cover.setdefault((p, t), []).append((m0, m0 + 64 * nk))
        for p, (M, N, K, nr, kr, br) in enumerate(probs):
            for t in range(((nr + 255) // 256) * ((K + 255) // 256)):
                segs = sorted(cover[(p, t)])
                assert segs[0][0] == 0 and segs[-1][1] == M and all(a[1] == b[0] for a, b in zip(segs, segs[1:])), (trial, p, t)


def test_torch_library_ops_registered():
    """The custom-operator layer registers its schemas without a GPU; calling one on CPU tensors is refused by PyTorch
    itself (no CPU kernel exists - the product has no fallback)."""
    import pytest
    import torch
    import ase_amd.ops  # noqa: F401
    for name in ('linear_act', 'linear_bwd_data', 'linear_bwd_weight', 'rms_update_normalize', 'rms_normalize', 'gae',
                 'masked_norm', 'gather_rows', 'disc_reward', 'enc_reward', 'normalize_rows', 'sample_latents', 'fused_adam_'):
        assert hasattr(torch.ops.ase_hip, name), name
    assert str(torch.ops.ase_hip.linear_act.default._schema) == 'ase_hip::linear_act(Tensor x, Tensor w, Tensor b, str act) -> Tensor'
    with pytest.raises(NotImplementedError):
        torch.ops.ase_hip.linear_act(torch.zeros(4, 8), torch.zeros(3, 8), torch.zeros(3), 'relu')


def test_every_entry_point_is_documented():
    """INTEGRATION.md (section C) names every function of the C ABI next to the reference statement it replaces - grouped
    families are written as `ase_hip_prog_create / begin / ...`."""
    doc = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    for name in _header_decls():
        stem, last = name.rsplit('_', 1)
        assert name in doc or (stem in doc and re.search(r'[/ ]\s*' + re.escape(last) + r'\b', doc)), f'{name} is not mentioned in INTEGRATION.md'


def test_scaler_entry_points_validate_on_the_host():
    """ase_hip_scaler_check / ase_hip_scaler_step reject null, empty, misaligned, badly typed and aliased operands before any
    launch (so this runs without a GPU), with the message naming the entry point."""
    lib = L.load()
    buf = (ctypes.c_double * 8)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    assert lib.ase_hip_scaler_check(None, 8, L.F16, p, None) == -1 and b'scaler_check' in lib.ase_hip_last_error()
    assert lib.ase_hip_scaler_check(p, 0, L.F16, p, None) == -1
    assert lib.ase_hip_scaler_check(p, 8, L.F32X3, p, None) == -1 and b'dtype' in lib.ase_hip_last_error()
    assert lib.ase_hip_scaler_check(ctypes.c_void_p(p.value + 1), 4, L.F16, p, None) == -1 and b'misaligned' in lib.ase_hip_last_error()
    assert lib.ase_hip_scaler_step(p, p, p, p, 8, None, None) == -1 and b'scaler_step' in lib.ase_hip_last_error()      # opt_eff aliases opt_state
    assert lib.ase_hip_scaler_step(None, p, p, p, 8, None, None) == -1
    assert lib.ase_hip_scaler_step(p, p, ctypes.c_void_p(p.value + 8), p, 0, None, None) == -1
    # ABI 7: the table form of the check and the fold of the records' counts
    assert lib.ase_hip_scaler_check_multi(None, 3, 64, p, None) == -1 and b'scaler_check_multi' in lib.ase_hip_last_error()
    assert lib.ase_hip_scaler_check_multi(p, 0, 64, p, None) == -1
    assert lib.ase_hip_scaler_check_multi(p, 3, 0, p, None) == -1 and b'wg_per_buf' in lib.ase_hip_last_error()
    assert lib.ase_hip_scaler_check_multi(p, 3, 64, None, None) == -1
    assert lib.ase_hip_scaler_fold(None, p, None) == -1 and b'scaler_fold' in lib.ase_hip_last_error()
    assert lib.ase_hip_scaler_fold(p, None, None) == -1
