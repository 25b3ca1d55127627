// The four random draws of one DrQV2Agent.update in ONE launch, bit-identical to the four ATen launches they replace.
//
// Reference order: torch.randint(0, 2*pad+1, (B,1,1,2), dtype=float32) twice (drqv2.py:34 from :241-242),
// then torch.empty((B,A)).normal_() twice (utils.py:135 from drqv2.py:183 and :211).  ATen draws them with
// distribution_elementwise_grid_stride_kernel (ATen/native/cuda/DistributionTemplates.h): thread i of a launch runs
// Philox4x32-10 with key = the generator's seed, subsequence = i, offset = the generator's offset; for the sizes of
// an update (numel <= 256 * grid cap) every thread keeps only the FIRST of its four values, and every launch advances
// the generator's offset by 4.  randint: x % range as float; normal_(0,1): the first of hiprand_normal4's Box-Muller
// pair products, times 1 plus 0.  The same hipRAND device functions are used here; compiled with -ffp-contract=on
// (build.py) they also round like torch's copy (tools/rng_flags_probe.py: 0 of 65,536 normals differ; hipcc's default
// contraction mode: 15 % differ in the last bit).  The host still proves it once per engine against torch itself
// before it trusts this path, and falls back to the two integer draws or to torch's own calls (engine.py:
// _rng_selftest).
#include "common.h"
#include <hiprand/hiprand_kernel.h>

namespace {

struct RngArgs {
  unsigned long long seed, offset;
  float* shift[2];   // [n_shift]
  float* noise[2];   // [n_noise]
  int n_shift, n_noise;
  unsigned range;
  int gs, gn;        // workgroups per shift / noise draw
};

__global__ __launch_bounds__(256) void rng_draws_kernel(RngArgs a) {
#pragma clang fp contract(off)
  int blk = blockIdx.x;
  const int draw = blk < 2 * a.gs ? blk / a.gs : 2 + (blk - 2 * a.gs) / a.gn;
  blk -= draw < 2 ? draw * a.gs : 2 * a.gs + (draw - 2) * a.gn;
  const int idx = blk * 256 + threadIdx.x;
  hiprandStatePhilox4_32_10_t st;
  if (draw < 2) {
    if (idx >= a.n_shift) return;
    hiprand_init(a.seed, (unsigned long long)idx, a.offset + 4ull * draw, &st);
    const uint4 r = hiprand4(&st);
    a.shift[draw][idx] = (float)(long long)(r.x % a.range);
  } else {
    if (idx >= a.n_noise) return;
    hiprand_init(a.seed, (unsigned long long)idx, a.offset + 4ull * draw, &st);
    const float4 z = hiprand_normal4(&st);
    a.noise[draw - 2][idx] = __fadd_rn(__fmul_rn(z.x, 1.0f), 0.0f);
  }
}

}  // namespace

extern "C" {

// shift_obs / shift_next: n_shift floats each (= 2*B), values in [0, range); noise_critic / noise_actor: n_noise
// floats each (= B*A).  (seed, offset) = the state of torch's CUDA generator BEFORE the draws; the caller advances the
// generator's offset by 16 (four launches' worth) -- or by 8 when n_noise == 0: only the two shift draws are made (the
// integer draws are exact by construction; the normal draws go through logf / sincosf, whose last bit depends on the
// device-library build, so a host whose torch rounds them differently keeps torch's own normal_ calls).  Sizes above 65536 elements per draw are refused (DRQ_EARG): ATen's
// launch geometry changes there and the caller keeps torch's own calls.
DRQ_API int drq_rng_draws(unsigned long long seed, unsigned long long offset, int n_shift, int n_noise, int range,
                          float* shift_obs, float* shift_next, float* noise_critic, float* noise_actor,
                          hipStream_t st) {
  if (!shift_obs || !shift_next || n_shift <= 0 || n_noise < 0 || range <= 0) return DRQ_EARG;
  if (n_noise > 0 && (!noise_critic || !noise_actor)) return DRQ_EARG;
  if (n_shift > 65536 || n_noise > 65536) return DRQ_EARG;
  RngArgs a{seed, offset, {shift_obs, shift_next}, {noise_critic, noise_actor}, n_shift, n_noise, (unsigned)range,
            (n_shift + 255) / 256, (n_noise + 255) / 256};
  hipLaunchKernelGGL(rng_draws_kernel, dim3(2 * a.gs + 2 * a.gn), dim3(256), 0, st, a);
  DRQ_LAUNCH_CHECK();
  return DRQ_OK;
}

}  // extern "C"
