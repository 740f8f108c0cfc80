// AMP observation production: one frame of the 140-float (per character: 1 + 6 + 3 + 3 + 6 J + D + 3 K)
// discriminator observation from the simulator state, pushed into the per-env history [N, S, F] whose flattened rows
// are the amp_obs the update path consumes.  HBM-light quaternion arithmetic: one lane per environment, the frame is
// staged through LDS so that both the history shift and the new slot are row-contiguous accesses.
// Follows env/tasks/humanoid_amp.py:248-266,280-316 and env/tasks/humanoid.py:523-552 (reference, /root/reference/ase).
#include "common.h"

namespace {

constexpr int kMaxJoints = 32;
constexpr int kEnvPerBlock = 64;

struct AmpObsArgs {
    const float *root_pos, *root_rot, *root_vel, *root_ang_vel, *dof_pos, *dof_vel, *key_pos;
    float* hist;            // [N, S, F]
    int N, D, K, J, S, F;
    int local_root, root_height, shift;
    int dof_off[kMaxJoints + 1];
};

struct V3 { float x, y, z; };
struct Q4 { float x, y, z, w; };

// v rotated by the unit quaternion q (xyzw):  v (2 w^2 - 1) + 2 w (u x v) + 2 u (u . v)
__device__ __forceinline__ V3 rot(const Q4& q, const V3& v) {
    const float a = 2.f * q.w * q.w - 1.f, d = 2.f * (q.x * v.x + q.y * v.y + q.z * v.z), w2 = 2.f * q.w;
    return V3{v.x * a + (q.y * v.z - q.z * v.y) * w2 + q.x * d,
              v.y * a + (q.z * v.x - q.x * v.z) * w2 + q.y * d,
              v.z * a + (q.x * v.y - q.y * v.x) * w2 + q.z * d};
}
__device__ __forceinline__ Q4 mul(const Q4& a, const Q4& b) {
    return Q4{a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x,
              a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}
__device__ __forceinline__ Q4 from_angle_axis(float angle, V3 ax) {
    const float n = fmaxf(sqrtf(ax.x * ax.x + ax.y * ax.y + ax.z * ax.z), 1e-9f);
    const float s = sinf(0.5f * angle) / n, c = cosf(0.5f * angle);
    Q4 q{ax.x * s, ax.y * s, ax.z * s, c};
    const float m = fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-9f);
    return Q4{q.x / m, q.y / m, q.z / m, q.w / m};
}
// tangent (rotated x axis) and normal (rotated z axis)
__device__ __forceinline__ void tan_norm(const Q4& q, float* o) {
    const V3 t = rot(q, V3{1.f, 0.f, 0.f}), n = rot(q, V3{0.f, 0.f, 1.f});
    o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = n.x; o[4] = n.y; o[5] = n.z;
}

__global__ __launch_bounds__(kEnvPerBlock) void amp_obs_kernel(AmpObsArgs a) {
    extern __shared__ float tile[];                       // [kEnvPerBlock][F + 1]
    const int F = a.F, pitch = F + 1;
    const int e0 = blockIdx.x * kEnvPerBlock, n = e0 + threadIdx.x;
    const int live = min(kEnvPerBlock, a.N - e0);
    if (n < a.N) {
        float* o = tile + threadIdx.x * pitch;
        const float* rp = a.root_pos + 3 * (int64_t)n;
        const float* rq = a.root_rot + 4 * (int64_t)n;
        const Q4 q{rq[0], rq[1], rq[2], rq[3]};
        const V3 d = rot(q, V3{1.f, 0.f, 0.f});
        const Q4 hq = from_angle_axis(-atan2f(d.y, d.x), V3{0.f, 0.f, 1.f});      // inverse heading rotation
        o[0] = a.root_height ? rp[2] : 0.f;
        tan_norm(a.local_root ? mul(hq, q) : q, o + 1);
        const float* v = a.root_vel + 3 * (int64_t)n;
        const float* w = a.root_ang_vel + 3 * (int64_t)n;
        const V3 lv = rot(hq, V3{v[0], v[1], v[2]}), lw = rot(hq, V3{w[0], w[1], w[2]});
        o[7] = lv.x; o[8] = lv.y; o[9] = lv.z; o[10] = lw.x; o[11] = lw.y; o[12] = lw.z;
        const float* dp = a.dof_pos + (int64_t)a.D * n;
        for (int j = 0; j < a.J; ++j) {
            const int b = a.dof_off[j], sz = a.dof_off[j + 1] - b;
            Q4 jq;
            if (sz == 3) {                               // exponential map -> quaternion
                const V3 e{dp[b], dp[b + 1], dp[b + 2]};
                const float len = sqrtf(e.x * e.x + e.y * e.y + e.z * e.z);
                float ang = atan2f(sinf(len), cosf(len));
                V3 ax{e.x / len, e.y / len, e.z / len};
                if (!(fabsf(ang) > 1e-5f)) { ang = 0.f; ax = V3{0.f, 0.f, 1.f}; }
                jq = from_angle_axis(ang, ax);
            } else {                                     // hinge about y
                jq = from_angle_axis(dp[b], V3{0.f, 1.f, 0.f});
            }
            tan_norm(jq, o + 13 + 6 * j);
        }
        const int od = 13 + 6 * a.J;
        const float* dv = a.dof_vel + (int64_t)a.D * n;
        for (int i = 0; i < a.D; ++i) o[od + i] = dv[i];
        const float* kp = a.key_pos + (int64_t)a.K * 3 * n;
        for (int k = 0; k < a.K; ++k) {
            const V3 l = rot(hq, V3{kp[3 * k] - rp[0], kp[3 * k + 1] - rp[1], kp[3 * k + 2] - rp[2]});
            o[od + a.D + 3 * k] = l.x; o[od + a.D + 3 * k + 1] = l.y; o[od + a.D + 3 * k + 2] = l.z;
        }
    }
    __syncthreads();
    // the new frame takes slot 0 (the shift kernel has already moved the past): threads walk feature columns of the
    // block's environments, so the stores are row-contiguous
    for (int e = 0; e < live; ++e) {
        float* h = a.hist + (int64_t)(e0 + e) * a.S * F;
        for (int f = threadIdx.x; f < F; f += kEnvPerBlock) h[f] = tile[e * pitch + f];
    }
}

// history slots move one step into the past (oldest dropped): one thread per (env, feature) walks its column from the
// oldest slot down, lanes cover consecutive features -> every access is a contiguous row segment
__global__ __launch_bounds__(256) void amp_hist_shift_kernel(float* __restrict__ hist, int64_t n_cols, int S, int F) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n_cols) return;
    const int64_t e = i / F;
    const int f = (int)(i - e * F);
    float* h = hist + e * S * F + f;
    for (int s = S - 2; s >= 0; --s) h[(int64_t)(s + 1) * F] = h[(int64_t)s * F];
}

// ---- motion clip sampler (utils/motion_lib.py:122-172,263-272,296-325; utils/torch_utils.py:7-28,94-118) -------------
struct MotionArgs {
    const float *gts, *grs, *lrs, *grvs, *gravs, *dvs;       // [frames, B, 3] [frames, B, 4] x2 [frames, 3] x2 [frames, D]
    const float *lengths, *dt;                               // per motion
    const int32_t *num_frames, *length_starts, *motion_ids;
    const float* times;
    float *root_pos, *root_rot, *dof_pos, *root_vel, *root_ang_vel, *dof_vel, *key_pos;
    int n, B, D, J, K;
    int dof_off[kMaxJoints + 1], dof_body[kMaxJoints], key_body[kMaxJoints];
};

__device__ __forceinline__ Q4 load_q(const float* p) { return Q4{p[0], p[1], p[2], p[3]}; }

__device__ __forceinline__ Q4 slerp(const Q4& a, Q4 b, float t) {
    float c = a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
    if (c < 0.f) b = Q4{-b.x, -b.y, -b.z, -b.w};
    c = fabsf(c);
    if (c >= 1.f) return a;
    const float ht = acosf(c), s = sqrtf(1.f - c * c);
    if (fabsf(s) < 0.001f) return Q4{0.5f * a.x + 0.5f * b.x, 0.5f * a.y + 0.5f * b.y, 0.5f * a.z + 0.5f * b.z, 0.5f * a.w + 0.5f * b.w};
    const float ra = sinf((1.f - t) * ht) / s, rb = sinf(t * ht) / s;
    return Q4{ra * a.x + rb * b.x, ra * a.y + rb * b.y, ra * a.z + rb * b.z, ra * a.w + rb * b.w};
}

__global__ __launch_bounds__(64) void motion_state_kernel(MotionArgs a) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= a.n) return;
    const int mid = a.motion_ids[i];
    const float t = a.times[i], len = a.lengths[mid], dt = a.dt[mid];
    const int nf = a.num_frames[mid];
    const float phase = fminf(fmaxf(t / len, 0.f), 1.f);
    const int i0 = (int)(phase * (float)(nf - 1));
    const int i1 = min(i0 + 1, nf - 1);
    const float blend = (t - (float)i0 * dt) / dt;
    const int64_t f0 = i0 + a.length_starts[mid], f1 = i1 + a.length_starts[mid];
    const int B = a.B;
    const float* p0 = a.gts + f0 * B * 3;
    const float* p1 = a.gts + f1 * B * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) a.root_pos[3 * (int64_t)i + c] = (1.f - blend) * p0[c] + blend * p1[c];
    const Q4 rr = slerp(load_q(a.grs + f0 * B * 4), load_q(a.grs + f1 * B * 4), blend);
    float* ro = a.root_rot + 4 * (int64_t)i;
    ro[0] = rr.x; ro[1] = rr.y; ro[2] = rr.z; ro[3] = rr.w;
    for (int k = 0; k < a.K; ++k) {
        const int b = a.key_body[k];
#pragma unroll
        for (int c = 0; c < 3; ++c)
            a.key_pos[((int64_t)i * a.K + k) * 3 + c] = (1.f - blend) * p0[b * 3 + c] + blend * p1[b * 3 + c];
    }
    float* dp = a.dof_pos + (int64_t)a.D * i;
    for (int j = 0; j < a.J; ++j) {
        const int b = a.dof_body[j], o = a.dof_off[j], sz = a.dof_off[j + 1] - o;
        const Q4 q = slerp(load_q(a.lrs + (f0 * B + b) * 4), load_q(a.lrs + (f1 * B + b) * 4), blend);
        // quaternion -> (angle, axis): below sin(theta/2) = 1e-5 (or NaN from w > 1) the rotation counts as none about z
        const float sn = sqrtf(1.f - q.w * q.w);
        float ang = 2.f * acosf(q.w);
        ang = atan2f(sinf(ang), cosf(ang));
        V3 ax{q.x / sn, q.y / sn, q.z / sn};
        if (!(fabsf(sn) > 1e-5f)) { ang = 0.f; ax = V3{0.f, 0.f, 1.f}; }
        if (sz == 3) {
            dp[o] = ang * ax.x; dp[o + 1] = ang * ax.y; dp[o + 2] = ang * ax.z;
        } else {
            const float th = ang * ax.y;                       // hinge joints turn about y
            dp[o] = atan2f(sinf(th), cosf(th));
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        a.root_vel[3 * (int64_t)i + c] = a.grvs[f0 * 3 + c];
        a.root_ang_vel[3 * (int64_t)i + c] = a.gravs[f0 * 3 + c];
    }
    for (int d = 0; d < a.D; ++d) a.dof_vel[(int64_t)a.D * i + d] = a.dvs[f0 * a.D + d];
}

}  // namespace

extern "C" int ase_hip_build_amp_obs(const float* root_pos, const float* root_rot, const float* root_vel,
                                     const float* root_ang_vel, const float* dof_pos, const float* dof_vel,
                                     const float* key_body_pos, int n_envs, int n_dof, int n_key,
                                     const int32_t* dof_offsets, int n_joints, int local_root_obs, int root_height_obs,
                                     float* hist, int n_steps, int shift, void* stream) {
    ASE_CHECK_ARG(root_pos && root_rot && root_vel && root_ang_vel && dof_pos && dof_vel && key_body_pos && hist && dof_offsets,
                  "build_amp_obs: null operand");
    ASE_CHECK_ARG(n_envs > 0 && n_dof > 0 && n_key >= 0 && n_steps >= 1 && n_joints >= 1 && n_joints <= kMaxJoints,
                  "build_amp_obs: bad sizes (envs %d, dofs %d, joints %d)", n_envs, n_dof, n_joints);
    AmpObsArgs a;
    a.root_pos = root_pos; a.root_rot = root_rot; a.root_vel = root_vel; a.root_ang_vel = root_ang_vel;
    a.dof_pos = dof_pos; a.dof_vel = dof_vel; a.key_pos = key_body_pos; a.hist = hist;
    a.N = n_envs; a.D = n_dof; a.K = n_key; a.J = n_joints; a.S = n_steps;
    a.local_root = local_root_obs; a.root_height = root_height_obs; a.shift = shift;
    for (int j = 0; j <= n_joints; ++j) {
        a.dof_off[j] = dof_offsets[j];
        if (j > 0) {
            const int sz = dof_offsets[j] - dof_offsets[j - 1];
            ASE_CHECK_ARG(sz == 1 || sz == 3, "build_amp_obs: joint %d has %d dofs (1 or 3 supported)", j - 1, sz);
        }
    }
    ASE_CHECK_ARG(dof_offsets[0] == 0 && dof_offsets[n_joints] == n_dof, "build_amp_obs: dof_offsets do not cover the dofs");
    a.F = 13 + 6 * n_joints + n_dof + 3 * n_key;
    const int lds = kEnvPerBlock * (a.F + 1) * (int)sizeof(float);
    ASE_CHECK_ARG(lds <= 64 * 1024, "build_amp_obs: frame of %d floats does not fit the staging tile", a.F);
    if (shift && n_steps > 1) {
        const int64_t cols = (int64_t)n_envs * a.F;
        ASE_LAUNCH(amp_hist_shift_kernel, dim3((unsigned)((cols + 255) / 256)), dim3(256), 0, (hipStream_t)stream, hist,
                           cols, n_steps, a.F);
    }
    ASE_LAUNCH(amp_obs_kernel, dim3((n_envs + kEnvPerBlock - 1) / kEnvPerBlock), dim3(kEnvPerBlock), lds,
                       (hipStream_t)stream, a);
    ASE_CHECK_LAUNCH("build_amp_obs");
    return ASE_OK;
}

extern "C" int ase_hip_motion_state(const float* gts, const float* grs, const float* lrs, const float* grvs,
                                    const float* gravs, const float* dvs, int n_bodies, const float* lengths,
                                    const int32_t* num_frames, const float* dt, const int32_t* length_starts,
                                    const int32_t* motion_ids, const float* times, int n, const int32_t* dof_body_ids,
                                    const int32_t* dof_offsets, int n_joints, const int32_t* key_body_ids, int n_key,
                                    float* root_pos, float* root_rot, float* dof_pos, float* root_vel,
                                    float* root_ang_vel, float* dof_vel, float* key_pos, void* stream) {
    ASE_CHECK_ARG(gts && grs && lrs && grvs && gravs && dvs && lengths && num_frames && dt && length_starts && motion_ids &&
                      times && dof_body_ids && dof_offsets && root_pos && root_rot && dof_pos && root_vel && root_ang_vel &&
                      dof_vel && (key_pos || n_key == 0) && (key_body_ids || n_key == 0),
                  "motion_state: null operand");
    ASE_CHECK_ARG(n > 0 && n_bodies > 0 && n_joints >= 1 && n_joints <= kMaxJoints && n_key >= 0 && n_key <= kMaxJoints,
                  "motion_state: bad sizes (n %d, bodies %d, joints %d, key bodies %d)", n, n_bodies, n_joints, n_key);
    MotionArgs a;
    a.gts = gts; a.grs = grs; a.lrs = lrs; a.grvs = grvs; a.gravs = gravs; a.dvs = dvs;
    a.lengths = lengths; a.dt = dt; a.num_frames = num_frames; a.length_starts = length_starts;
    a.motion_ids = motion_ids; a.times = times;
    a.root_pos = root_pos; a.root_rot = root_rot; a.dof_pos = dof_pos; a.root_vel = root_vel;
    a.root_ang_vel = root_ang_vel; a.dof_vel = dof_vel; a.key_pos = key_pos;
    a.n = n; a.B = n_bodies; a.J = n_joints; a.K = n_key; a.D = dof_offsets[n_joints];
    for (int j = 0; j <= n_joints; ++j) a.dof_off[j] = dof_offsets[j];
    for (int j = 0; j < n_joints; ++j) {
        const int sz = dof_offsets[j + 1] - dof_offsets[j];
        ASE_CHECK_ARG((sz == 1 || sz == 3) && dof_body_ids[j] >= 0 && dof_body_ids[j] < n_bodies,
                      "motion_state: joint %d: %d dofs on body %d", j, sz, dof_body_ids[j]);
        a.dof_body[j] = dof_body_ids[j];
    }
    for (int k = 0; k < n_key; ++k) {
        ASE_CHECK_ARG(key_body_ids[k] >= 0 && key_body_ids[k] < n_bodies, "motion_state: key body %d out of range", key_body_ids[k]);
        a.key_body[k] = key_body_ids[k];
    }
    ASE_LAUNCH(motion_state_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, a);
    ASE_CHECK_LAUNCH("motion_state");
    return ASE_OK;
}
