"""Tensor-level wrapper over the C ABI: extracts pointers / leading dimensions from torch tensors
that live in HBM and enqueues the HIP kernels on torch's current stream.

PyTorch is used here only for device memory and streams.  All arithmetic of the update path
happens inside ``libase_hip.so``.  There is no CPU implementation in the product; the engine's
host logic is exercised on CPU by ``tests/emu_backend.py`` (test infrastructure with the same
method names).
"""
import ctypes as C

import torch

from . import lib as L


_HOSTFN = C.CFUNCTYPE(None, C.c_void_p)


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _ld(t):
    return 0 if t is None else int(t.stride(0))


def _code(dtype):
    if dtype == torch.bfloat16:
        return L.BF16
    if dtype == torch.float32:
        return L.F32
    if dtype == torch.float16:
        return L.F16
    raise L.AseHipError(f"unsupported storage dtype {dtype}")


class HipBackend:
    name = "hip"

    def __init__(self, device=None, x3=False):
        # x3: f32-stored GEMM operands are multiplied as three 16-bit MFMAs on a hi/lo split - True: bf16 parts (ASE_F32X3, any
        # operand range), 'f16': half parts of operands scaled by 2^ea / 2^eb (ASE_F32H3, gemm_nt only; the caller passes the
        # exponents of a launch as x3_exps=(ea, eb), see f32h_t in csrc/common.h)
        self.x3 = x3 if x3 == 'f16' else bool(x3)
        self._recording, self._host_keep, self._host_error = None, {}, None
        self.tn_workspace = True     # grouped weight gradients: partial tiles + reduce kernel (False: f32 atomics into G)
        if not torch.cuda.is_available():
            raise L.AseHipError("HipBackend needs a ROCm GPU (torch.cuda.is_available() is False); "
                                "the update path has no CPU fallback")
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        self.lib = L.get()
        self._head_scratch = torch.zeros(L.PPO_SCRATCH, dtype=torch.float64, device=self.device)   # (zeroed: it ends in a ticket word)

    # ------------------------------------------------------------------ helpers
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def zero_(self, t):
        assert t.is_contiguous()
        L.check(self.lib.ase_hip_memset(_ptr(t), 0, t.numel() * t.element_size(), self._stream()), "memset")

    def copy_(self, dst, src):
        assert dst.is_contiguous() and src.is_contiguous() and dst.dtype == src.dtype and dst.numel() == src.numel()
        L.check(self.lib.ase_hip_memcpy(_ptr(dst), _ptr(src), dst.numel() * dst.element_size(), self._stream()), "memcpy")

    # ------------------------------------------------------------------ streams: fork / join, launch programs
    def mark(self):
        """Event at the current position of torch's current stream -> id (fork point / branch completion)."""
        ev = C.c_int(0)
        L.check(self.lib.ase_hip_mark(self._stream(), C.byref(ev)), "mark")
        return ev.value

    def wait(self, ev):
        """torch's current stream waits for the event."""
        L.check(self.lib.ase_hip_wait(self._stream(), int(ev)), "wait")

    def prog_create(self):
        p = C.c_void_p()
        L.check(self.lib.ase_hip_prog_create(C.byref(p)), "prog_create")
        return p

    def prog_begin(self, prog):
        L.check(self.lib.ase_hip_prog_begin(prog), "prog_begin")
        self._recording = prog.value
        self._host_keep.setdefault(prog.value, [])

    def prog_end(self, prog):
        self._recording = None
        L.check(self.lib.ase_hip_prog_end(prog), "prog_end")

    def host_call(self, fn):
        """A host-side operation at this position of the launch sequence (a collective, a torch op between kernels): run
        now, or - while a launch program records - on every replay (ase_hip_prog_host).  fn enqueues its own GPU work."""
        if self._recording is None:
            fn()
            return

        def trampoline(_arg, fn=fn):
            try:
                fn()
            except BaseException as e:           # an exception cannot cross the C frame: keep it for the caller of prog_launch
                self._host_error = e
        cb = _HOSTFN(trampoline)
        self._host_keep[self._recording].append(cb)          # the program calls through this object on every replay
        L.check(self.lib.ase_hip_prog_host(C.cast(cb, C.c_void_p), None), "prog_host")

    def prog_launch(self, prog):
        L.check(self.lib.ase_hip_prog_launch(prog), "prog_launch")
        if self._host_error is not None:
            e, self._host_error = self._host_error, None
            raise e

    def prog_size(self, prog):
        return self.lib.ase_hip_prog_size(prog)

    def prog_destroy(self, prog):
        self.lib.ase_hip_prog_destroy(prog)
        self._host_keep.pop(prog.value, None)

    def _gemm_code(self, dtype, exps=None):
        c = _code(dtype)
        if self.x3 and c == L.F32:
            if self.x3 == 'f16' and exps is not None:
                return L.F32H3 | (int(exps[0]) << 8) | (int(exps[1]) << 16)
            return L.F32X3
        return c

    # ------------------------------------------------------------------ GEMMs
    def gemm_nt(self, A, B, Cm, M, N, K, bias=None, aux=None, aux_mode=L.AUX_NONE, colsum=None, colsum_n=0,
                act=L.ACT_NONE, alpha=1.0, aux_split=0, aux_delta=0, mask_out=None, x3_exps=None, alpha_dev=None):
        """alpha_dev (all `*_dev` / `dyn` arguments of this class): a scale RECORD on the device, f32 {factor, overflow count} - an
        entry of the dynamic loss scale's table (UpdateEngine.scale_tab: S, 1 / S, 1 / S^2, 1).  The launch multiplies its scale by the
        factor when it RUNS and adds to the count when an element it stored overflowed (include/ase_hip.h, ABI 7)."""
        dt = self._gemm_code(A.dtype, x3_exps)
        assert B.dtype == A.dtype
        if aux_mode == L.AUX_RELU_BITS:
            assert aux.dtype == torch.int32
        else:
            assert aux is None or aux.dtype == A.dtype
        # twin output: the ReLU bit matrix (int32 words), or for the smooth activations the pre-activation in the storage type
        assert mask_out is None or mask_out.dtype == (A.dtype if act >= L.ACT_SILU else torch.int32)
        out_f32 = int(Cm.dtype == torch.float32 and A.dtype != torch.float32)
        L.check(self.lib.ase_hip_gemm_nt(_ptr(A), _ld(A), _ptr(B), _ld(B), _ptr(Cm), _ld(Cm), _ptr(bias), _ptr(aux),
                                         _ld(aux), int(aux_split), int(aux_delta), _ptr(colsum), int(colsum_n),
                                         _ptr(mask_out), _ld(mask_out), M, N, K, act, aux_mode, out_f32,
                                         float(alpha), _ptr(alpha_dev), dt, self._stream()), "gemm_nt")

    def gemm_tn(self, A, B, G, M, N, K, n_real, k_real, split_src, split_dst, alpha=1.0, gbias=None, bias_rows=0, alpha_dev=None):
        L.check(self.lib.ase_hip_gemm_tn(_ptr(A), _ld(A), _ptr(B), _ld(B), _ptr(G), _ptr(gbias), int(bias_rows), M, N, K,
                                         n_real, k_real,
                                         split_src, split_dst, float(alpha), _ptr(alpha_dev), self._gemm_code(A.dtype), self._stream()),
                "gemm_tn")

    # grouped weight gradients: one launch for every (eligible) dense layer of a step
    def grouped_tn_ok(self, dtype, M, n_real, K, bias_rows):
        """Problems the phased bf16 kernel takes: whole 64-row K-tiles.  Narrow outputs (heads, style MLP: N or K <= 64)
        waste most of a 256 x 256 tile's MFMAs, but a work item costs its HBM traffic either way: measured (scripts/lab,
        LAB_TNG_N) the six narrow problems of a step add 67 us to the grouped launch against 156 us as seven launches of the
        128 x 128 kernel."""
        # (every problem: narrow outputs - heads, style MLP, N or K <= 64 - waste most of a 256 x 256 tile's MFMAs, but a work
        #  item costs its HBM traffic either way; measured: the six narrow problems of a step add 67 us to the grouped launch
        #  against 156 us as seven launches of the 128 x 128 kernel)
        return dtype in (torch.bfloat16, torch.float16) and M % 64 == 0 and bias_rows % 64 == 0

    def make_tn_plan(self, problems, target_wg=0, alpha_dev=None):
        """problems: [(A, B, G, gbias|None, bias_rows, M, N, K, n_real, k_real, split_src, split_dst, alpha)] -> plan
        (device tables + the tensors they point to, kept alive)."""
        import struct
        n = len(problems)
        tab = (C.c_int64 * (16 * n))()
        for i, (A, B, G, gb, br, M, N, K, nr, kr, ss, sd, alpha) in enumerate(problems):
            assert A.dtype in (torch.bfloat16, torch.float16) and B.dtype == A.dtype and G.dtype == torch.float32
            row = [A.data_ptr(), _ld(A), B.data_ptr(), _ld(B), G.data_ptr(), 0 if gb is None else gb.data_ptr(), int(br),
                   M, N, K, nr, kr, ss, sd, struct.unpack('<i', struct.pack('<f', float(alpha)))[0], 0]
            for j, v in enumerate(row):
                tab[16 * i + j] = int(v)
        max_work = 8192
        work = (C.c_int32 * (4 * max_work))()
        red = (C.c_int32 * (4 * max_work))()
        n_work, n_red = C.c_int(0), C.c_int(0)
        L.check(self.lib.ase_hip_gemm_tn_grouped_plan(tab, n, int(target_wg), work, max_work, C.byref(n_work), red, max_work,
                                                      C.byref(n_red)), "gemm_tn_grouped_plan")
        nw, nr = n_work.value, n_red.value
        dev_tab = torch.tensor(list(tab), dtype=torch.int64, device=self.device)
        dev_work = torch.tensor(list(work[:4 * nw]), dtype=torch.int32, device=self.device)
        dev_red = torch.tensor(list(red[:4 * nr]), dtype=torch.int32, device=self.device)
        # partial-sum workspace of this launch (plans of different branches run side by side: one each)
        ws = torch.empty(nw * L.TN_SLAB, dtype=torch.float32, device=self.device) if self.tn_workspace else None
        return {'problems': dev_tab, 'work': dev_work, 'n_work': nw, 'red': dev_red, 'n_red': nr, 'ws': ws, 'keep': problems,
                'dtype': _code(problems[0][0].dtype), 'alpha_dev': alpha_dev}

    def gemm_tn_grouped(self, plan):
        L.check(self.lib.ase_hip_gemm_tn_grouped(_ptr(plan['problems']), _ptr(plan['work']), plan['n_work'], _ptr(plan['red']),
                                                 plan['n_red'], _ptr(plan['ws']), _ptr(plan.get('alpha_dev')), plan['dtype'],
                                                 self._stream()),
                "gemm_tn_grouped")

    def refresh_shadow(self, W, Ws, Wts, split_src, split_dst, x3_exp=None):
        """x3_exp (f32 shadows only): write them in the packed half-split format of ASE_F32H3 - W * 2^x3_exp as [8 hi | 8 lo] halves
        per group of 8 elements, the B operand of gemm_nt(..., x3_exps=(ea, x3_exp)) under x3 = 'f16'."""
        n, k = W.shape
        ref = Ws if Ws is not None else Wts
        code = _code(ref.dtype)
        if x3_exp is not None:
            assert ref.dtype == torch.float32
            code = L.F32H3 | (int(x3_exp) << 16)
        L.check(self.lib.ase_hip_refresh_shadow(_ptr(W), n, k, _ptr(Ws), _ld(Ws), _ptr(Wts), _ld(Wts), split_src,
                                                split_dst, code, self._stream()), "refresh_shadow")

    def refresh_shadow_multi(self, desc, items, dtype):
        """desc: device int64 [n, 12] pointer table (see ase_hip.h); items: the same tensors (kept alive by the caller)."""
        L.check(self.lib.ase_hip_refresh_shadow_multi(_ptr(desc), desc.shape[0], _code(dtype), self._stream()),
                "refresh_shadow_multi")

    def apply_multi(self, desc, items, dtype, opt_state, acc):
        """Fused optimizer step + shadow refresh of every layer (desc: device int64 [n, 24], see ase_hip.h)."""
        L.check(self.lib.ase_hip_apply_multi(_ptr(desc), desc.shape[0], _ptr(opt_state), _ptr(acc), _code(dtype),
                                             self._stream()), "apply_multi")

    def gather_multi(self, desc, items, idx, remap, M):
        L.check(self.lib.ase_hip_gather_multi(_ptr(desc), desc.shape[0], _ptr(idx), remap[0], remap[1], M,
                                              self._stream()), "gather_multi")

    # ------------------------------------------------------------------ observation side (N2)
    def build_amp_obs(self, root_pos, root_rot, root_vel, root_ang_vel, dof_pos, dof_vel, key_body_pos, dof_offsets,
                      local_root_obs, root_height_obs, hist, shift=True):
        """One AMP-observation frame per env pushed into hist [N, S, F] (env/tasks/humanoid_amp.py:248-316)."""
        n, S, F = hist.shape
        offs = (C.c_int32 * len(dof_offsets))(*[int(x) for x in dof_offsets])
        for t in (root_pos, root_rot, root_vel, root_ang_vel, dof_pos, dof_vel, key_body_pos, hist):
            assert t.dtype == torch.float32 and t.is_contiguous()
        L.check(self.lib.ase_hip_build_amp_obs(_ptr(root_pos), _ptr(root_rot), _ptr(root_vel), _ptr(root_ang_vel), _ptr(dof_pos),
                                               _ptr(dof_vel), _ptr(key_body_pos), n, dof_pos.shape[1], key_body_pos.shape[1], offs,
                                               len(dof_offsets) - 1, int(local_root_obs), int(root_height_obs), _ptr(hist), S,
                                               int(shift), self._stream()), "build_amp_obs")

    def motion_state(self, clips, motion_ids, times):
        """MotionLib.get_motion_state (utils/motion_lib.py:122-172) on device clip tensors.  clips: dict with gts, grs, lrs,
        grvs, gravs, dvs (f32), lengths, dt (f32), num_frames, length_starts (int32) on this device and the python lists
        dof_body_ids, dof_offsets, key_body_ids.  Returns (root_pos, root_rot, dof_pos, root_vel, root_ang_vel, dof_vel, key_pos)."""
        n, B = motion_ids.shape[0], clips['gts'].shape[1]
        D, J, K = clips['dof_offsets'][-1], len(clips['dof_body_ids']), len(clips['key_body_ids'])
        ia = lambda xs: (C.c_int32 * len(xs))(*[int(x) for x in xs])
        f = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=self.device)
        out = (f(n, 3), f(n, 4), f(n, D), f(n, 3), f(n, 3), f(n, D), f(n, K, 3))
        assert motion_ids.dtype == torch.int32 and times.dtype == torch.float32
        L.check(self.lib.ase_hip_motion_state(_ptr(clips['gts']), _ptr(clips['grs']), _ptr(clips['lrs']), _ptr(clips['grvs']),
                                              _ptr(clips['gravs']), _ptr(clips['dvs']), B, _ptr(clips['lengths']),
                                              _ptr(clips['num_frames']), _ptr(clips['dt']), _ptr(clips['length_starts']),
                                              _ptr(motion_ids), _ptr(times), n, ia(clips['dof_body_ids']), ia(clips['dof_offsets']), J,
                                              ia(clips['key_body_ids']), K, *[_ptr(o) for o in out], self._stream()), "motion_state")
        return out

    # ------------------------------------------------------------------ normaliser / gather
    def rms_moments(self, src, D, idx, remap, M, state, sums):
        L.check(self.lib.ase_hip_rms_moments(_ptr(src), _ld(src), D, _ptr(idx), remap[0], remap[1], M, _ptr(state),
                                             _ptr(sums), self._stream()), "rms_moments")

    @staticmethod
    def _stream_arrays(streams):
        n = len(streams)
        srcs = (C.c_void_p * n)(*[s[0].data_ptr() for s in streams])
        lds = (C.c_int64 * n)(*[int(s[0].stride(0)) for s in streams])
        idxs = (C.c_void_p * n)(*[None if s[1] is None else s[1].data_ptr() for s in streams])
        rh = (C.c_int * n)(*[int(s[2][0]) for s in streams])
        rn = (C.c_int * n)(*[int(s[2][1]) for s in streams])
        return n, srcs, lds, idxs, rh, rn

    def rms_moments_multi(self, streams, D, M, state, sums_list):
        """streams: [(src, idx, remap)] (<= 4, same width D and row count M); sums_list: one f64 [2 D] buffer per stream."""
        n, srcs, lds, idxs, rh, rn = self._stream_arrays(streams)
        sums = (C.c_void_p * n)(*[t.data_ptr() for t in sums_list])
        L.check(self.lib.ase_hip_rms_moments_multi(srcs, lds, idxs, rh, rn, sums, n, D, M, _ptr(state), self._stream()),
                "rms_moments_multi")

    def rms_normalize_multi(self, streams, D, M, means, stds, outs):
        n, srcs, lds, idxs, rh, rn = self._stream_arrays(streams)
        mp = (C.c_void_p * n)(*[t.data_ptr() for t in means])
        sp = (C.c_void_p * n)(*[t.data_ptr() for t in stds])
        op = (C.c_void_p * n)(*[t.data_ptr() for t in outs])
        lo = (C.c_int64 * n)(*[int(t.stride(0)) for t in outs])
        L.check(self.lib.ase_hip_rms_normalize_multi(srcs, lds, idxs, rh, rn, mp, sp, op, lo, n, D, M, _code(outs[0].dtype),
                                                     self._stream()), "rms_normalize_multi")

    def rms_finalize(self, state, D, sums, count, n_streams, mean_out, std_out):
        counts = (C.c_int32 * max(n_streams, 1))(*([int(count)] * max(n_streams, 1)))
        L.check(self.lib.ase_hip_rms_finalize(_ptr(state), D, _ptr(sums), counts, n_streams, _ptr(mean_out),
                                              _ptr(std_out), self._stream()), "rms_finalize")

    def rms_normalize(self, src, D, idx, remap, M, mean, std, outs):
        outs = list(outs) + [None] * (3 - len(outs))
        dt = _code(outs[0].dtype)
        L.check(self.lib.ase_hip_rms_normalize(_ptr(src), _ld(src), D, _ptr(idx), remap[0], remap[1], M, _ptr(mean),
                                               _ptr(std), _ptr(outs[0]), _ld(outs[0]), _ptr(outs[1]), _ld(outs[1]),
                                               _ptr(outs[2]), _ld(outs[2]), dt, self._stream()), "rms_normalize")

    def rms_unnormalize(self, state, x, y):
        L.check(self.lib.ase_hip_rms_unnormalize(_ptr(state), _ptr(x), _ptr(y), x.numel(), self._stream()),
                "rms_unnormalize")

    def gather_rows(self, src, D, idx, remap, M, dst):
        L.check(self.lib.ase_hip_gather_rows(_ptr(src), _ld(src), D, _ptr(idx), remap[0], remap[1], M, _ptr(dst),
                                             _ld(dst), _code(dst.dtype), self._stream()), "gather_rows")

    # ------------------------------------------------------------------ heads
    def reduce_sum(self, x, n, square, acc, slot):
        L.check(self.lib.ase_hip_reduce_sum(_ptr(x), n, int(square), _ptr(acc), slot, self._stream()), "reduce_sum")

    def ppo_head(self, mu, value, mb, new_z, logstd, d_mu, d_value, db_mu, db_value, acc, M, m_global, act_dim,
                 z_dim, masked, div_on, mu_tanh, clip_value, e_clip, critic_coef, bounds_coef, div_coef, div_tar,
                 mu_out=None, grad_scale=1.0, dyn=None):
        L.check(self.lib.ase_hip_ppo_head(
            _ptr(mu), _ld(mu), _ptr(value), _ld(value), _ptr(mb['actions']), _ptr(mb['mu']), _ptr(mb['sigma']),
            _ptr(mb['old_logp_actions']), _ptr(mb['advantages']), _ptr(mb.get('old_values')), _ptr(mb['returns']),
            _ptr(mb.get('rand_action_mask')), _ptr(mb.get('ase_latents')), _ptr(new_z), _ptr(logstd),
            _ptr(d_mu), _ld(d_mu), _ptr(d_value), _ld(d_value), _ptr(db_mu), _ptr(db_value), _ptr(mu_out), _ptr(acc),
            _ptr(self._head_scratch),
            M, m_global, act_dim, z_dim, int(masked), int(div_on), int(mu_tanh), int(clip_value),
            float(e_clip), float(critic_coef), float(bounds_coef), float(div_coef), float(div_tar), float(grad_scale),
            _ptr(dyn), _code(d_mu.dtype), self._stream()), "ppo_head")

    def disc_head(self, logit, d_logit, db_logit, acc, amb, amb_global, disc_coef, grad_scale=1.0, dyn=None):
        L.check(self.lib.ase_hip_disc_head(_ptr(logit), _ld(logit), _ptr(d_logit), _ld(d_logit), _ptr(db_logit),
                                           _ptr(acc), amb, amb_global, float(disc_coef), float(grad_scale), _ptr(dyn),
                                           _code(d_logit.dtype), self._stream()), "disc_head")

    def enc_head(self, e, z, d_e, db_enc, enc_out, acc, amb, amb_global, z_dim, enc_coef, grad_scale=1.0, dyn=None):
        L.check(self.lib.ase_hip_enc_head(_ptr(e), _ld(e), _ptr(z), _ld(z), _ptr(d_e), _ld(d_e), _ptr(db_enc),
                                          _ptr(enc_out), _ptr(acc), amb, amb_global, z_dim, float(enc_coef), float(grad_scale),
                                          _ptr(dyn), _code(d_e.dtype), self._stream()), "enc_head")

    def enc_gp_seed(self, e, z, u, rows, z_dim, scale=1.0):
        """u[:rows, :z_dim] = scale * d enc_err / d e (learning/ase_agent.py:431-434; e f32 pre-normalisation output)."""
        assert e.dtype == torch.float32 and z.dtype == torch.float32
        L.check(self.lib.ase_hip_enc_gp_seed(_ptr(e), _ld(e), _ptr(z), _ld(z), _ptr(u), _ld(u), rows, z_dim, float(scale),
                                             _code(u.dtype), self._stream()), "enc_gp_seed")

    def enc_gp_back(self, e, z, du, d_e, db_enc, rows, z_dim, grad_scale=1.0, dyn=None):
        """d_e[:rows, :z_dim] += (d u / d e) du, the bias gradient follows the stored values."""
        assert e.dtype == torch.float32 and z.dtype == torch.float32 and du.dtype == torch.float32
        L.check(self.lib.ase_hip_enc_gp_back(_ptr(e), _ld(e), _ptr(z), _ld(z), _ptr(du), _ld(du), _ptr(d_e), _ld(d_e),
                                             _ptr(db_enc), rows, z_dim, float(grad_scale), _ptr(dyn), _code(d_e.dtype),
                                             self._stream()), "enc_gp_back")

    def gp_seed(self, h, w, g, rows, width, scale=1.0, act=L.ACT_RELU):
        L.check(self.lib.ase_hip_gp_seed(_ptr(h), _ld(h), _ptr(w), _ptr(g), _ld(g), rows, width, float(scale), int(act),
                                         _code(h.dtype), self._stream()), "gp_seed")

    def gp_second(self, twin, g, dg, dz, rows, width, act):
        """dz += act'' / act'^2 * g * dg (second-order term of the gradient penalty's backward, smooth activations)."""
        L.check(self.lib.ase_hip_gp_second(_ptr(twin), _ld(twin), _ptr(g), _ld(g), _ptr(dg), _ld(dg), _ptr(dz), _ld(dz), rows,
                                           width, int(act), _code(dz.dtype), self._stream()), "gp_second")

    def colsum(self, x, rows, cols, out, scale=1.0):
        assert x.dtype == torch.float32 and out.dtype == torch.float32
        L.check(self.lib.ase_hip_colsum(_ptr(x), _ld(x), rows, cols, float(scale), _ptr(out), self._stream()), "colsum")

    def sqnorm(self, x, rows, cols, acc, slot, scale=1.0, dyn=None):
        L.check(self.lib.ase_hip_sqnorm(_ptr(x), _ld(x), rows, cols, _ptr(acc), slot, float(scale), _ptr(dyn), _code(x.dtype),
                                        self._stream()),
                "sqnorm")

    def finalize_scalars(self, acc, out, m_global, amb_global, masked, has_disc, has_enc, has_div, c, opt_state=None, kl_threshold=0.0):
        L.check(self.lib.ase_hip_finalize_scalars(
            _ptr(acc), _ptr(out), m_global, amb_global, int(masked), int(has_disc), int(has_enc), int(has_div),
            float(c['critic_coef']), float(c['entropy_coef']), float(c.get('bounds_loss_coef') or 0.0), float(c.get('disc_coef', 0)),
            float(c.get('disc_logit_reg', 0)), float(c.get('disc_grad_penalty', 0)), float(c.get('disc_weight_decay', 0)),
            float(c.get('enc_coef', 0)), float(c.get('enc_weight_decay', 0)), float(c.get('amp_diversity_bonus', 0)),
            float(c.get('enc_grad_penalty', 0)), _ptr(opt_state), float(kl_threshold), self._stream()), "finalize_scalars")

    # ------------------------------------------------------------------ optimizer
    def begin_step(self, opt_state, acc, zero2=None, rng_bump=None):
        L.check(self.lib.ase_hip_begin_step(_ptr(opt_state), _ptr(acc), 0 if acc is None else acc.numel(), _ptr(zero2),
                                            0 if zero2 is None else zero2.numel(), _ptr(rng_bump), self._stream()),
                "begin_step")

    def adam(self, w, g, m, v, opt_state):
        L.check(self.lib.ase_hip_adam(_ptr(w), _ptr(g), _ptr(m), _ptr(v), w.numel(), _ptr(opt_state), self._stream()),
                "adam")

    def clip_scale(self, g, acc, slot, max_norm):
        L.check(self.lib.ase_hip_clip_scale(_ptr(g), g.numel(), _ptr(acc[slot:]), float(max_norm), self._stream()), "clip_scale")

    def axpy(self, g, w, c):
        L.check(self.lib.ase_hip_axpy(_ptr(g), _ptr(w), g.numel(), float(c), self._stream()), "axpy")

    def scaler_check(self, buf, scaler):
        """GradScaler's found_inf test over one buffer of the scaled backward (any view of contiguous storage)."""
        assert buf.is_contiguous()
        L.check(self.lib.ase_hip_scaler_check(_ptr(buf), buf.numel(), _code(buf.dtype), _ptr(scaler), self._stream()),
                "scaler_check")

    def scaler_check_multi(self, bufs, scaler, table=None):
        """The same test over a list of buffers in ONE launch; `table` = what make_check_table(bufs) returned (built once)."""
        if table is None:
            table = self.make_check_table(bufs)
        L.check(self.lib.ase_hip_scaler_check_multi(_ptr(table['rows']), table['n'], table['wg'], _ptr(scaler), self._stream()),
                "scaler_check_multi")

    def make_check_table(self, bufs):
        rows = []
        for t in bufs:
            assert t.is_contiguous() and t.numel() > 0
            rows.append([t.data_ptr(), t.numel(), _code(t.dtype)])
        big = max(t.numel() * t.element_size() for t in bufs)
        wg = max(1, min(1024, (big + 16383) // 16384))             # four 16-byte loads per thread of the largest buffer
        return {'rows': torch.tensor(rows, dtype=torch.int64, device=self.device), 'n': len(rows), 'wg': int(wg), 'keep': list(bufs)}

    def scaler_fold(self, scaler, scale_tab):
        """Overflow counts of the scale records -> scaler[found] (in front of the ranks' exchange of that flag)."""
        L.check(self.lib.ase_hip_scaler_fold(_ptr(scaler), _ptr(scale_tab), self._stream()), "scaler_fold")

    def scaler_step(self, scaler, opt_state, opt_eff, grads, scale_tab=None):
        """GradScaler.step's decision - a found overflow zeroes the gradient and hands the optimizer launch the identity step - and
        (scale_tab given) GradScaler.update(): backoff / growth of the device-resident scale and its table, every step."""
        L.check(self.lib.ase_hip_scaler_step(_ptr(scaler), _ptr(opt_state), _ptr(opt_eff), _ptr(grads), grads.numel(),
                                             _ptr(scale_tab), self._stream()), "scaler_step")

    # ------------------------------------------------------------------ rollout tail
    def disc_reward(self, logit, r, n, scale):
        L.check(self.lib.ase_hip_disc_reward(_ptr(logit), _ld(logit), _ptr(r), n, float(scale), self._stream()),
                "disc_reward")

    def enc_reward(self, e, z, r, n, z_dim, scale):
        L.check(self.lib.ase_hip_enc_reward(_ptr(e), _ld(e), _ptr(z), _ld(z), _ptr(r), n, z_dim, float(scale),
                                            self._stream()), "enc_reward")

    def gae(self, dones, values, next_values, r_task, r_disc, r_enc, w_task, w_disc, w_enc, gamma, tau, advs, returns,
            H, N):
        L.check(self.lib.ase_hip_gae(_ptr(dones), _ptr(values), _ptr(next_values), _ptr(r_task), _ptr(r_disc),
                                     _ptr(r_enc), float(w_task), float(w_disc), float(w_enc), float(gamma), float(tau),
                                     _ptr(advs), _ptr(returns), H, N, self._stream()), "gae")

    def adv_norm(self, returns, values, mask, adv, acc3, n, normalize, phase):
        L.check(self.lib.ase_hip_adv_norm(_ptr(returns), _ptr(values), _ptr(mask), _ptr(adv), _ptr(acc3), n,
                                          int(normalize), phase, self._stream()), "adv_norm")

    def ring_store(self, src, D, idx, remap, n, dst, size, head):
        L.check(self.lib.ase_hip_ring_store(_ptr(src), _ld(src), D, _ptr(idx), remap[0], remap[1], n, _ptr(dst), size,
                                            head, self._stream()), "ring_store")

    def normalize_rows(self, x, y, n, dim):
        L.check(self.lib.ase_hip_normalize_rows(_ptr(x), _ld(x), _ptr(y), _ld(y), n, dim, self._stream()), "normalize_rows")

    def sample_actions(self, mu, logstd, rand_probs, rng_state, mu_out, sigma_out, actions, neglogp, rand_mask, n, act_dim,
                       mu_tanh=False):
        L.check(self.lib.ase_hip_sample_actions(_ptr(mu), _ld(mu), _ptr(logstd), _ptr(rand_probs), _ptr(rng_state), _ptr(mu_out),
                                                _ptr(sigma_out), _ptr(actions), _ptr(neglogp), _ptr(rand_mask), n, act_dim,
                                                int(mu_tanh), self._stream()), "sample_actions")

    def sample_latents(self, z, rows, dim, rng_state, row_offset=0, advance=True, z2=None):
        L.check(self.lib.ase_hip_sample_latents(_ptr(z), rows, dim, _ptr(rng_state), int(row_offset), int(advance), _ptr(z2),
                                                _ld(z2), _code(z2.dtype) if z2 is not None else 0, self._stream()),
                "sample_latents")
