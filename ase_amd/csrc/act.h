// Activation functions of the dense layers (rl_games activations_factory names: relu, tanh, sigmoid, elu, selu, swish,
// gelu, softplus, None - learning/ase_network_builder.py:162, learning/amp_network_builder.py:98-117), with the first and
// second derivative as functions of the PRE-activation z: the data-gradient epilogues multiply by act'(z) (ASE_AUX_PREACT,
// z kept by the forward launch in the layer's twin buffer) and the gradient penalty's double backward
// (learning/amp_agent.py:453-459) needs act''(z).  Same formulas as torch.nn.functional (ELU alpha 1, exact-erf GELU,
// Softplus beta 1 / threshold 20).
#pragma once
#include "../../include/ase_hip.h"

__device__ __forceinline__ float act_sigmoid(float z) { return 1.f / (1.f + expf(-z)); }

__device__ __forceinline__ float act_apply(int act, float z) {
    switch (act) {
        case ASE_ACT_RELU: return fmaxf(z, 0.f);
        case ASE_ACT_TANH: return tanhf(z);
        case ASE_ACT_SILU: return z * act_sigmoid(z);
        case ASE_ACT_ELU: return z > 0.f ? z : expm1f(z);
        case ASE_ACT_GELU: return 0.5f * z * (1.f + erff(z * 0.70710678118654752440f));
        case ASE_ACT_SIGMOID: return act_sigmoid(z);
        case ASE_ACT_SELU: return 1.0507009873554804934f * (z > 0.f ? z : 1.6732632423543772848f * expm1f(z));
        case ASE_ACT_SOFTPLUS: return z > 20.f ? z : log1pf(expf(z));
        default: return z;
    }
}

__device__ __forceinline__ float act_grad(int act, float z) {
    switch (act) {
        case ASE_ACT_RELU: return z > 0.f ? 1.f : 0.f;
        case ASE_ACT_TANH: { const float t = tanhf(z); return 1.f - t * t; }
        case ASE_ACT_SILU: { const float s = act_sigmoid(z); return s * (1.f + z * (1.f - s)); }
        case ASE_ACT_ELU: return z > 0.f ? 1.f : expf(z);
        case ASE_ACT_GELU:
            return 0.5f * (1.f + erff(z * 0.70710678118654752440f)) + z * 0.39894228040143267794f * expf(-0.5f * z * z);
        case ASE_ACT_SIGMOID: { const float s = act_sigmoid(z); return s * (1.f - s); }
        case ASE_ACT_SELU: return 1.0507009873554804934f * (z > 0.f ? 1.f : 1.6732632423543772848f * expf(z));
        case ASE_ACT_SOFTPLUS: return z > 20.f ? 1.f : act_sigmoid(z);
        default: return 1.f;
    }
}

__device__ __forceinline__ float act_grad2(int act, float z) {
    switch (act) {
        case ASE_ACT_TANH: { const float t = tanhf(z); return -2.f * t * (1.f - t * t); }
        case ASE_ACT_SILU: { const float s = act_sigmoid(z), ds = s * (1.f - s); return ds * (2.f + z * (1.f - 2.f * s)); }
        case ASE_ACT_ELU: return z > 0.f ? 0.f : expf(z);
        case ASE_ACT_GELU: return 0.39894228040143267794f * expf(-0.5f * z * z) * (2.f - z * z);
        case ASE_ACT_SIGMOID: { const float s = act_sigmoid(z); return s * (1.f - s) * (1.f - 2.f * s); }
        case ASE_ACT_SELU: return z > 0.f ? 0.f : 1.0507009873554804934f * 1.6732632423543772848f * expf(z);
        case ASE_ACT_SOFTPLUS: { if (z > 20.f) return 0.f; const float s = act_sigmoid(z); return s * (1.f - s); }
        default: return 0.f;      // identity, ReLU (almost everywhere)
    }
}
