// gsr_optim.hip -- the optimiser step of a refinement iteration.
//
// GauSTAR steps every parameter group with torch.optim.Adam (gaustar_scene/sugar_optimizer.py:87, :99-101: Adam over
// `_points`, SH coefficients, densities, scales, quaternions, loose-bind offsets; lr = 0 defaults, eps = 1e-15, per-group
// learning rates).  At 491 520 Gaussians that is 24 M parameters = 675 MB of traffic per step (read p, g, m, v; write p,
// m, v); PyTorch's multi-tensor kernels move it at about 2 TB/s on this part (0.32 ms per iteration, a quarter of an
// iteration built from this package's fused ops).  adam_kernel is the same update as torch/optim/adam.py::_single_tensor_adam
// (no weight decay, no amsgrad, not maximising), one 16-byte access per lane per array, nothing else: HBM-bound.
//     m += (g - m) (1 - beta1)                      exp_avg.lerp_(grad, 1 - beta1)
//     v  = v beta2 + g g (1 - beta2)                exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
//     p -= step_size * m / (sqrt(v) / bc2s + eps)   step_size = lr / (1 - beta1^t), bc2s = sqrt(1 - beta2^t)
#include "gsr_internal.h"

namespace gsr {

__device__ __forceinline__ void adam_update(float& p, float g, float& m, float& v, float step_size, float one_minus_b1,
                                            float b2, float one_minus_b2, float eps, float bc2s)
{
#pragma clang fp contract(off)
    m = m + (g - m) * one_minus_b1;
    v = v * b2 + (g * g) * one_minus_b2;
    const float denom = sqrtf(v) / bc2s + eps;
    p = p - step_size * (m / denom);
}

__global__ void __launch_bounds__(256)
adam_kernel(long long n, float* __restrict__ param, const float* __restrict__ grad, float* __restrict__ exp_avg,
            float* __restrict__ exp_avg_sq, float step_size, float one_minus_b1, float b2, float one_minus_b2, float eps,
            float bc2s)
{
    const long long n4 = n >> 2;
    const long long stride = (long long)gridDim.x * blockDim.x;
    float4* p4 = reinterpret_cast<float4*>(param);
    const float4* g4 = reinterpret_cast<const float4*>(grad);
    float4* m4 = reinterpret_cast<float4*>(exp_avg);
    float4* v4 = reinterpret_cast<float4*>(exp_avg_sq);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 p = p4[i], m = m4[i], v = v4[i];
        const float4 g = g4[i];
        adam_update(p.x, g.x, m.x, v.x, step_size, one_minus_b1, b2, one_minus_b2, eps, bc2s);
        adam_update(p.y, g.y, m.y, v.y, step_size, one_minus_b1, b2, one_minus_b2, eps, bc2s);
        adam_update(p.z, g.z, m.z, v.z, step_size, one_minus_b1, b2, one_minus_b2, eps, bc2s);
        adam_update(p.w, g.w, m.w, v.w, step_size, one_minus_b1, b2, one_minus_b2, eps, bc2s);
        p4[i] = p; m4[i] = m; v4[i] = v;
    }
    // up to three trailing elements
    const long long t = (n4 << 2) + (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) adam_update(param[t], grad[t], exp_avg[t], exp_avg_sq[t], step_size, one_minus_b1, b2, one_minus_b2, eps, bc2s);
}

// The same update over up to ADAM_BATCH tensors in ONE launch: a refinement iteration steps eight parameter tensors, five of
// them small enough that their own launch is all overhead (5 - 8 us each for 0.5 - 6 MB).  The tensor table travels in
// the kernel arguments; workgroup -> tensor by a scan of at most 16 first-workgroup indices; inside a tensor, the
// single-tensor kernel's loop.
__global__ void __launch_bounds__(256)
adam_multi_kernel(AdamBatch b, float one_minus_b1, float b2, float one_minus_b2, float eps, float bc2s)
{
    int ti = 0;
#pragma unroll
    for (int i = 1; i < ADAM_BATCH; i++)
        if (i < b.count && blockIdx.x >= b.t[i].block0) ti = i;
    const AdamTensor t = b.t[ti];
    const unsigned nblk = (ti + 1 < b.count ? b.t[ti + 1].block0 : b.blocks) - t.block0;
    const unsigned blk = blockIdx.x - t.block0;
    const long long n4 = t.n >> 2;
    const long long stride = (long long)nblk * blockDim.x;
    float4* p4 = reinterpret_cast<float4*>(t.param);
    const float4* g4 = reinterpret_cast<const float4*>(t.grad);
    float4* m4 = reinterpret_cast<float4*>(t.exp_avg);
    float4* v4 = reinterpret_cast<float4*>(t.exp_avg_sq);
    for (long long i = (long long)blk * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 p = p4[i], m = m4[i], v = v4[i];
        const float4 g = g4[i];
        adam_update(p.x, g.x, m.x, v.x, t.step_size, one_minus_b1, b2, one_minus_b2, eps, bc2s);
        adam_update(p.y, g.y, m.y, v.y, t.step_size, one_minus_b1, b2, one_minus_b2, eps, bc2s);
        adam_update(p.z, g.z, m.z, v.z, t.step_size, one_minus_b1, b2, one_minus_b2, eps, bc2s);
        adam_update(p.w, g.w, m.w, v.w, t.step_size, one_minus_b1, b2, one_minus_b2, eps, bc2s);
        p4[i] = p; m4[i] = m; v4[i] = v;
    }
    const long long tail = (n4 << 2) + (long long)blk * blockDim.x + threadIdx.x;
    if (tail < t.n)
        adam_update(t.param[tail], t.grad[tail], t.exp_avg[tail], t.exp_avg_sq[tail], t.step_size, one_minus_b1, b2, one_minus_b2,
                    eps, bc2s);
}

void launch_adam_multi(const AdamBatch& b, float one_minus_b1, float b2, float one_minus_b2, float eps, float bc2s, hipStream_t st)
{
    adam_multi_kernel<<<b.blocks, 256, 0, st>>>(b, one_minus_b1, b2, one_minus_b2, eps, bc2s);
}

void launch_adam(long long n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float step_size,
                 float one_minus_b1, float b2, float one_minus_b2, float eps, float bc2s, hipStream_t st)
{
    // one 16-byte element per thread up to 256 k workgroups (grid caps between 2 k and 16 k workgroups all measured
    // slower or no faster: 5.3 - 6.1 TB/s, run-to-run spread included)
    const long long n4 = n >> 2;
    long long blocks = (n4 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 256 * 1024) blocks = 256 * 1024;
    adam_kernel<<<(unsigned)blocks, 256, 0, st>>>(n, param, grad, exp_avg, exp_avg_sq, step_size, one_minus_b1, b2, one_minus_b2,
                                                  eps, bc2s);
}

}  // namespace gsr
