// Attention pooling over variable-length bags / graphs (gfx950).
//
// Replaces, for a whole batch of bags in ONE launch (the reference runs one bag
// per Python call):
//   utils_g_mil.py:72-97   AttentionMIL_teacher: scores -> softmax over the bag ->
//                          class-space pooling of patch logits, patch/bag softmax
//   utils_g_mil.py:32-33   AttentionMIL: feature-space pooling z = sum a*h
//   05_train_gnns.py:205-213  GraphMIL multi-head pool, mean over heads
//
// HBM-bound: every h row and t row is read exactly once (forward) through
// coalesced wave-wide loads; the softmax over the ragged bag is an online
// (running max / running sum) reduction held in registers per wave and merged
// across the 4 waves of the workgroup through LDS, so no second pass over h.
// One 256-thread workgroup per bag; waves stride the bag's instances.
#include "common.h"

namespace {

// Waves per bag (round 3; four before): a wave handles one instance row per memory latency, so a 196-node graph on four
// waves was 49 dependent round trips (121 us forward at 256 graphs = 1.1 TB/s of an 8 TB/s memory).  The backward kernel
// (61 VGPRs) runs sixteen waves; the forward kernel needs 217 VGPRs (class-space branch), so eight (512 threads).
constexpr int NWAVE = 8;                   // forward
constexpr int NTHR = NWAVE * 64;
constexpr int NWAVE_B = 16;                // backward
constexpr int NTHR_B = NWAVE_B * 64;
constexpr int MAX_HEADS = 4;   // heads per launch (host loops over groups of 4)
constexpr int MAX_C = 16;

struct PoolArgs {
  const float* h; const float* t; const float* w3; const float* b3; const float* W4; const float* b4;
  const int64_t* offsets;
  int B, H, A, heads, C, max_bag;
  int head0, heads_total;       // this launch handles heads [head0, head0+heads)
  float* att; float* z; float* patch_logits; float* patch_probs; float* bag_logits; float* bag_probs;
  int z_accumulate;
};

// dynamic LDS carve (floats):
//   w3s[heads*A] | W4s[C*H] | sc[max_bag*heads] | wm[NWAVE*heads] | wl[NWAVE*heads] |
//   wP[NWAVE*MAX_C] | wz[NWAVE*heads*H] | Ms[4] | Ls[4] | fws[4*NWAVE]
// JA > 0: A <= 64 * JA -- the attention-hidden values of a row (all heads) are fetched up front together with the
// h values, so that a row costs ONE memory latency instead of one per head and 64-column slice (the generic loop,
// JA = 0, was latency-bound at an eighth of the HBM rate on GraphMIL's 4 heads x 128).
template <int JH, int JA>
__global__ __launch_bounds__(NTHR) void attn_pool_fwd_kernel(PoolArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int H = a.H, A = a.A, NH = a.heads, C = a.C;
  float* w3s = smem;
  float* W4s = w3s + NH * A;
  float* sc = W4s + (a.W4 ? C * H : 0);
  float* wm = sc + a.max_bag * NH;
  float* wl = wm + NWAVE * NH;
  float* wP = wl + NWAVE * NH;
  float* wz = wP + NWAVE * MAX_C;
  float* Ms = wz + (size_t)NWAVE * NH * H;       // merged running max / sum per head and the waves' rescale factors
  float* Ls = Ms + MAX_HEADS;
  float* fws = Ls + MAX_HEADS;                   // [MAX_HEADS][NWAVE]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x;
  const int64_t lo = a.offsets[b], hi = a.offsets[b + 1];
  const int nb = (int)(hi - lo);

  for (int i = tid; i < NH * A; i += NTHR) w3s[i] = a.w3[(size_t)a.head0 * A + i];
  if (a.W4)
    for (int i = tid; i < C * H; i += NTHR) W4s[i] = a.W4[i];
  __syncthreads();

  float m[MAX_HEADS], l[MAX_HEADS], zacc[MAX_HEADS][JH], accP[MAX_C];
#pragma unroll
  for (int k = 0; k < MAX_HEADS; ++k) {
    m[k] = -INFINITY; l[k] = 0.f;
#pragma unroll
    for (int j = 0; j < JH; ++j) zacc[k][j] = 0.f;
  }
#pragma unroll
  for (int c = 0; c < MAX_C; ++c) accP[c] = 0.f;

  const int At = a.heads_total * A;
  // A row's values are requested ONE ROW AHEAD of its arithmetic (round 3): a wave's rows were a chain of "load, then ~200
  // instructions" with nothing in flight meanwhile -- 25 dependent memory round trips for a 196-node graph on eight waves.
  // (The requests follow the first use of the current row's registers -- sched_barrier -- so that the wait hipcc puts in front
  //  of that use cannot include them.)
  float hnext[JH];
  float tnext[MAX_HEADS][JA > 0 ? JA : 1];
  auto request_row = [&](int n) {
    const int nn = n < nb ? n : (nb > 0 ? nb - 1 : 0);       // past the end: a valid row, never used
    const float* hrow = a.h + (size_t)(lo + nn) * H;
    const float* trow = a.t + (size_t)(lo + nn) * At + (size_t)a.head0 * A;
    // (unconditional loads at clamped, always valid addresses, masked where they are USED: a load under `if (col < H)` is a
    //  branch around it and a select right behind it a wait -- hipcc then fetched a row's values two at a time, five
    //  dependent round trips per row)
#pragma unroll
    for (int j = 0; j < JH; ++j) {
      const int col = lane + 64 * j;
      hnext[j] = hrow[min(col, H - 1)];                     // (columns >= H: a duplicate, masked where it is used)
    }
    if (JA > 0) {
#pragma unroll
      for (int k = 0; k < MAX_HEADS; ++k)
#pragma unroll
        for (int i = 0; i < (JA > 0 ? JA : 1); ++i) {
          const int j = lane + 64 * i;
          tnext[k][i] = trow[min(k, NH - 1) * A + min(j, A - 1)];
        }
    }
  };
  if (wave < nb) request_row(wave);
  for (int n = wave; n < nb; n += NWAVE) {
    const float* trow = a.t + (size_t)(lo + n) * At + (size_t)a.head0 * A;
    float hreg[JH];
    float treg[MAX_HEADS][JA > 0 ? JA : 1];
#pragma unroll
    for (int j = 0; j < JH; ++j) hreg[j] = hnext[j];
#pragma unroll
    for (int k = 0; k < MAX_HEADS; ++k)
#pragma unroll
      for (int i = 0; i < (JA > 0 ? JA : 1); ++i) treg[k][i] = tnext[k][i];
    __builtin_amdgcn_sched_barrier(0);
    request_row(n + NWAVE);
    __builtin_amdgcn_sched_barrier(0);
    // class-space branch: P[n,c] = W4[c,:] . h[n,:] + b4[c]
    float Pn[MAX_C];
    if (a.W4) {
#pragma unroll
      for (int c = 0; c < MAX_C; ++c) {
        if (c < C) {
          float p = 0.f;
#pragma unroll
          for (int j = 0; j < JH; ++j) {
            const int col = lane + 64 * j;
            if (col < H) p += hreg[j] * W4s[c * H + col];
          }
          Pn[c] = wave_sum(p) + a.b4[c];
        } else Pn[c] = 0.f;
      }
      if (a.patch_logits && lane < C) {
        float v = 0.f;
#pragma unroll
        for (int c = 0; c < MAX_C; ++c) if (lane == c) v = Pn[c];
        a.patch_logits[(size_t)(lo + n) * C + lane] = v;
      }
      if (a.patch_probs) {
        float mx = -INFINITY;
#pragma unroll
        for (int c = 0; c < MAX_C; ++c) if (c < C) mx = fmaxf(mx, Pn[c]);
        float se = 0.f, mine = 0.f;
#pragma unroll
        for (int c = 0; c < MAX_C; ++c) if (c < C) { const float e = expf(Pn[c] - mx); se += e; if (lane == c) mine = e; }
        if (lane < C) a.patch_probs[(size_t)(lo + n) * C + lane] = mine / se;
      }
    }
#pragma unroll
    for (int k = 0; k < MAX_HEADS; ++k) {
      if (k < NH) {
        float p = 0.f;
        if (JA > 0) {
#pragma unroll
          for (int i = 0; i < (JA > 0 ? JA : 1); ++i) {
            const int j = lane + 64 * i;
            if (j < A) p += treg[k][i] * w3s[k * A + j];
          }
        } else {
          for (int j = lane; j < A; j += 64) p += trow[k * A + j] * w3s[k * A + j];
        }
        const float s = wave_sum(p) + a.b3[a.head0 + k];
        if (lane == 0) sc[n * NH + k] = s;
        const float mn = fmaxf(m[k], s);
        const float f = expf(m[k] - mn);   // exp(-inf) = 0 on the first instance
        const float e = expf(s - mn);
        l[k] = l[k] * f + e;
        m[k] = mn;
#pragma unroll
        for (int j = 0; j < JH; ++j) zacc[k][j] = zacc[k][j] * f + e * hreg[j];
        if (k == 0 && a.W4) {
#pragma unroll
          for (int c = 0; c < MAX_C; ++c) accP[c] = accP[c] * f + e * Pn[c];
        }
      }
    }
  }

  // ---- merge the waves
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < MAX_HEADS; ++k) if (k < NH) { wm[wave * NH + k] = m[k]; wl[wave * NH + k] = l[k]; }
#pragma unroll
    for (int c = 0; c < MAX_C; ++c) wP[wave * MAX_C + c] = accP[c];
  }
  if (a.z) {
#pragma unroll
    for (int k = 0; k < MAX_HEADS; ++k) if (k < NH) {
#pragma unroll
      for (int j = 0; j < JH; ++j) {
        const int col = lane + 64 * j;
        if (col < H) wz[(wave * NH + k) * H + col] = zacc[k][j];
      }
    }
  }
  __syncthreads();
  // merge factors, once per head (LDS): with sixteen waves a per-thread copy of fw[heads][waves] spilled to scratch
  if (tid < NH) {
    const int k = tid;
    float M = -INFINITY, L = 0.f;
    for (int w = 0; w < NWAVE; ++w) M = fmaxf(M, wm[w * NH + k]);
    for (int w = 0; w < NWAVE; ++w) {
      const float mw = wm[w * NH + k];
      const float f = (mw == -INFINITY) ? 0.f : expf(mw - M);
      fws[k * NWAVE + w] = f;
      L += wl[w * NH + k] * f;
    }
    Ms[k] = M; Ls[k] = L;
  }
  __syncthreads();
  // attention weights
  for (int i = tid; i < nb * NH; i += NTHR) {
    const int n = i / NH, k = i - n * NH;
    a.att[(size_t)(lo + n) * a.heads_total + a.head0 + k] = expf(sc[i] - Ms[k]) / Ls[k];
  }
  // pooled features: mean over ALL heads of sum_n a*h
  if (a.z) {
    const float inv_heads = 1.f / (float)a.heads_total;
    for (int col = tid; col < H; col += NTHR) {
      float zs = 0.f;
      for (int k = 0; k < NH; ++k) if (nb > 0) {
        float zk = 0.f;
#pragma unroll
        for (int w = 0; w < NWAVE; ++w) zk += wz[(w * NH + k) * H + col] * fws[k * NWAVE + w];
        zs += zk / Ls[k];
      }
      zs *= inv_heads;
      float* zp = a.z + (size_t)b * H + col;
      *zp = a.z_accumulate ? *zp + zs : zs;
    }
  }
  if (a.W4 && a.bag_logits && tid == 0) {
    float bl[MAX_C], mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < MAX_C; ++c) {
      bl[c] = 0.f;
      if (c < C) {
        if (nb > 0) {
#pragma unroll
          for (int w = 0; w < NWAVE; ++w) bl[c] += wP[w * MAX_C + c] * fws[w];
          bl[c] /= Ls[0];
        }
        a.bag_logits[(size_t)b * C + c] = bl[c];
        mx = fmaxf(mx, bl[c]);
      }
    }
    if (a.bag_probs) {
      float se = 0.f;
#pragma unroll
      for (int c = 0; c < MAX_C; ++c) if (c < C) { bl[c] = expf(bl[c] - mx); se += bl[c]; }
#pragma unroll
      for (int c = 0; c < MAX_C; ++c) if (c < C) a.bag_probs[(size_t)b * C + c] = bl[c] / se;
    }
  }
}

// ---- GraphMIL form (no class-space branch, H <= 128, A <= 128, all heads in one launch) in TWO passes (round 3):
// the online-softmax kernel above spends ~350 instructions per instance row on ONE wave (four wave reductions, eight
// exponentials, the running rescale of the pooled sums) and a 196-node graph is 25 such rows in sequence per wave, two waves
// per SIMD: instruction-bound at 61 us for 256 graphs.  Here
//   pass 1: a wave per row computes only the heads' scores (attention-hidden row . w3) -> LDS;
//   softmax over the bag per head in LDS (wave k owns head k), attention weights written out, a[n] = sum_k att[n,k] kept;
//   pass 2: thread (g, col) sums a[n] * h[n, col] over the rows n = g (mod 4) -- coalesced row reads, no reductions across
//           lanes -- and the four partial sums meet in LDS in a fixed order.
// LDS (floats): w3s[NH*A] | sc[max_bag*NH] | asum[max_bag] | zp[4][H] | Ms[4] | Ls[4]
__global__ __launch_bounds__(NTHR) void attn_pool_fwd2_kernel(PoolArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int H = a.H, A = a.A, NH = a.heads;
  float* w3s = smem;
  float* sc = w3s + NH * A;
  float* asum = sc + a.max_bag * NH;
  float* zp = asum + a.max_bag;
  float* Ms = zp + 4 * H;
  float* Ls = Ms + MAX_HEADS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x;
  const int64_t lo = a.offsets[b], hi = a.offsets[b + 1];
  const int nb = (int)(hi - lo);
  for (int i = tid; i < NH * A; i += NTHR) w3s[i] = a.w3[i];
  __syncthreads();

  // ---- pass 1: scores.  Lane l holds columns l, l + 64 of every head's slice; the next row's values are requested before
  //      this row's reductions (unconditional clamped loads, masked where used: see attn_pool_fwd_kernel)
  const int At = NH * A;
  // (two rows per wave and iteration: their reductions interleave)
  float tn[2][MAX_HEADS][2];
  auto request = [&](int n) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int nr = n + r * NWAVE;
      const int nn = nr < nb ? nr : (nb > 0 ? nb - 1 : 0);
      const float* trow = a.t + (size_t)(lo + nn) * At;
#pragma unroll
      for (int k = 0; k < MAX_HEADS; ++k)
#pragma unroll
        for (int i = 0; i < 2; ++i) tn[r][k][i] = trow[min(k, NH - 1) * A + min(lane + 64 * i, A - 1)];
    }
  };
  if (wave < nb) request(wave);
  for (int n = wave; n < nb; n += 2 * NWAVE) {
    float tr[2][MAX_HEADS][2];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int k = 0; k < MAX_HEADS; ++k) { tr[r][k][0] = tn[r][k][0]; tr[r][k][1] = tn[r][k][1]; }
    __builtin_amdgcn_sched_barrier(0);
    request(n + 2 * NWAVE);
    __builtin_amdgcn_sched_barrier(0);
    float p[2][MAX_HEADS];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int k = 0; k < MAX_HEADS; ++k) {
        float q = 0.f;
        if (k < NH) {
          if (lane < A) q += tr[r][k][0] * w3s[k * A + lane];
          if (lane + 64 < A) q += tr[r][k][1] * w3s[k * A + lane + 64];
        }
        p[r][k] = q;
      }
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int k = 0; k < MAX_HEADS; ++k) p[r][k] = wave_sum(p[r][k]);
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int nr = n + r * NWAVE;
      if (nr < nb && lane == 0) {
#pragma unroll
        for (int k = 0; k < MAX_HEADS; ++k) if (k < NH) sc[nr * NH + k] = p[r][k] + a.b3[k];
      }
    }
  }
  __syncthreads();
  // ---- softmax over the bag, head k on wave k (fixed lane-strided order, then a fixed reduction tree)
  if (wave < NH) {
    const int k = wave;
    float mx = -INFINITY;
    for (int n = lane; n < nb; n += 64) mx = fmaxf(mx, sc[n * NH + k]);
    mx = wave_max(mx);
    float se = 0.f;
    for (int n = lane; n < nb; n += 64) se += expf(sc[n * NH + k] - mx);
    se = wave_sum(se);
    if (lane == 0) { Ms[k] = mx; Ls[k] = se; }
  }
  __syncthreads();
  for (int i = tid; i < nb * NH; i += NTHR) {
    const int k = i % NH;
    const float v = expf(sc[i] - Ms[k]) / Ls[k];
    sc[i] = v;
    a.att[(size_t)lo * NH + i] = v;
  }
  __syncthreads();
  for (int n = tid; n < nb; n += NTHR) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < MAX_HEADS; ++k) if (k < NH) s += sc[n * NH + k];
    asum[n] = s;
  }
  __syncthreads();
  // ---- pass 2: z[col] = (1 / heads) sum_n a[n] h[n, col]
  if (a.z) {
    const int col = tid & 127, g = tid >> 7;               // NTHR = 512: four row groups
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (col < H) {
      const float* hp = a.h + (size_t)lo * H + col;
      int n = g;
      for (; n + 12 < nb; n += 16) {                       // four rows of this group in flight
        const float h0 = hp[(size_t)n * H], h1 = hp[(size_t)(n + 4) * H], h2 = hp[(size_t)(n + 8) * H], h3 = hp[(size_t)(n + 12) * H];
        s0 += asum[n] * h0; s1 += asum[n + 4] * h1; s2 += asum[n + 8] * h2; s3 += asum[n + 12] * h3;
      }
      for (; n < nb; n += 4) s0 += asum[n] * hp[(size_t)n * H];
      zp[g * H + col] = (s0 + s1) + (s2 + s3);
    }
    __syncthreads();
    if (tid < H) {
      const float zs = ((zp[tid] + zp[H + tid]) + (zp[2 * H + tid] + zp[3 * H + tid])) / (float)a.heads_total;
      a.z[(size_t)b * H + tid] = zs;
    }
  }
}

struct PoolBwdArgs {
  const float* h; const float* t; const float* att; const float* P; const float* w3; const float* W4;
  const int64_t* offsets;
  int B, H, A, heads, C, max_bag;
  const float* d_bag_logits; const float* d_z;
  float* d_h; int accumulate_dh; float* d_u; float* d_s; float* d_P;
  float* psum;        // FAST only: [B][2 * heads * A + heads] per-bag sums over the bag's rows of d_u | d_s * t | d_s, or NULL
};

// LDS (floats): w3s[heads*A] | W4s[C*H] | dzs[H] | da[max_bag*heads] | red[NWAVE_B*heads] | psum: wred[NWAVE_B][2*heads*A + heads]
// FAST: H <= 128 and A <= 128 -- the h / attention-hidden values of a row are fetched up front (two per lane and head)
// instead of through runtime-length loops of dependent loads (as attn_pool_fwd_kernel<JH, 2>)
template <bool FAST>
__global__ __launch_bounds__(NTHR_B) void attn_pool_bwd_kernel(PoolBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int H = a.H, A = a.A, NH = a.heads, C = a.C;
  float* w3s = smem;
  float* W4s = w3s + NH * A;
  float* dzs = W4s + (a.W4 ? C * H : 0);
  float* da = dzs + H;
  float* red = da + a.max_bag * NH;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x;
  const int64_t lo = a.offsets[b], hi = a.offsets[b + 1];
  const int nb = (int)(hi - lo);
  const float inv_heads = 1.f / (float)NH;

  for (int i = tid; i < NH * A; i += NTHR_B) w3s[i] = a.w3[i];
  if (a.W4)
    for (int i = tid; i < C * H; i += NTHR_B) W4s[i] = a.W4[i];
  for (int i = tid; i < H; i += NTHR_B) dzs[i] = a.d_z ? a.d_z[(size_t)b * H + i] : 0.f;
  float dL[MAX_C];
#pragma unroll
  for (int c = 0; c < MAX_C; ++c) dL[c] = (a.d_bag_logits && c < C) ? a.d_bag_logits[(size_t)b * C + c] : 0.f;
  __syncthreads();

  // pass 1: da[n,k] = dL . P[n] (head 0 of the teacher form) + (1/heads) dz . h[n];  dot_k = sum_n a da
  float dot[MAX_HEADS];
#pragma unroll
  for (int k = 0; k < MAX_HEADS; ++k) dot[k] = 0.f;
  // (GraphMIL form, FAST: two rows per wave and iteration -- their loads are in flight together and their reductions
  //  interleave; rows n, n + NWAVE_B)
  int n1 = wave;
  if (FAST && !a.W4 && a.d_z) {
    for (; n1 + NWAVE_B < nb; n1 += 2 * NWAVE_B) {
      float hv[2][2], av[2][MAX_HEADS];
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int n = n1 + r * NWAVE_B;
        const float* hrow = a.h + (size_t)(lo + n) * H;
        hv[r][0] = hrow[min(lane, H - 1)];
        hv[r][1] = hrow[min(lane + 64, H - 1)];
#pragma unroll
        for (int k = 0; k < MAX_HEADS; ++k) av[r][k] = a.att[(size_t)(lo + n) * NH + min(k, NH - 1)];
      }
      float p[2];
#pragma unroll
      for (int r = 0; r < 2; ++r)
        p[r] = (lane < H ? hv[r][0] * dzs[lane] : 0.f) + (lane + 64 < H ? hv[r][1] * dzs[lane + 64] : 0.f);
#pragma unroll
      for (int r = 0; r < 2; ++r) p[r] = wave_sum(p[r]) * inv_heads;
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int n = n1 + r * NWAVE_B;
#pragma unroll
        for (int k = 0; k < MAX_HEADS; ++k) if (k < NH) {
          if (lane == 0) da[n * NH + k] = p[r];
          dot[k] += av[r][k] * p[r];
        }
      }
    }
  }
  for (int n = n1; n < nb; n += NWAVE_B) {
    float base = 0.f;
    if (a.d_z) {
      const float* hrow = a.h + (size_t)(lo + n) * H;
      float p = 0.f;
      if (FAST) {
        const float h0 = lane < H ? hrow[lane] : 0.f, h1 = lane + 64 < H ? hrow[lane + 64] : 0.f;
        p = (lane < H ? h0 * dzs[lane] : 0.f) + (lane + 64 < H ? h1 * dzs[lane + 64] : 0.f);
      } else {
        for (int j = lane; j < H; j += 64) p += hrow[j] * dzs[j];
      }
      base = wave_sum(p) * inv_heads;
    }
    float cls = 0.f;
    if (a.W4 && a.d_bag_logits) {
      float p = 0.f;
#pragma unroll
      for (int c = 0; c < MAX_C; ++c) if (c < C && lane == c) p = dL[c] * a.P[(size_t)(lo + n) * C + c];
      cls = wave_sum(p);
    }
#pragma unroll
    for (int k = 0; k < MAX_HEADS; ++k) if (k < NH) {
      const float d = base + (k == 0 ? cls : 0.f);
      if (lane == 0) da[n * NH + k] = d;
      dot[k] += a.att[(size_t)(lo + n) * NH + k] * d;
    }
  }
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < MAX_HEADS; ++k) if (k < NH) red[wave * NH + k] = dot[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < MAX_HEADS; ++k) if (k < NH) {
    float d = 0.f;
#pragma unroll
    for (int w = 0; w < NWAVE_B; ++w) d += red[w * NH + k];          // fixed order
    dot[k] = d;
  }

  // pass 2
  const int At = NH * A;
  // (psum) the column sums the parameter gradients need -- db2 = sum_n d_u, dw3 = sum_n d_s * t, db3 = sum_n d_s -- are
  // taken HERE, where d_u and t are in registers: a lane owns columns lane, lane + 64 of every head, a wave its rows
  float su[MAX_HEADS][2], sw[MAX_HEADS][2], ssum[MAX_HEADS];
#pragma unroll
  for (int k = 0; k < MAX_HEADS; ++k) { su[k][0] = su[k][1] = sw[k][0] = sw[k][1] = 0.f; ssum[k] = 0.f; }
  for (int n = wave; n < nb; n += NWAVE_B) {
    float an[MAX_HEADS], ds[MAX_HEADS];
#pragma unroll
    for (int k = 0; k < MAX_HEADS; ++k) {
      an[k] = 0.f; ds[k] = 0.f;
      if (k < NH) {
        an[k] = a.att[(size_t)(lo + n) * NH + k];
        ds[k] = an[k] * (da[n * NH + k] - dot[k]);
        if (a.d_s && lane == 0) a.d_s[(size_t)(lo + n) * NH + k] = ds[k];
      }
    }
    if (a.d_u) {
      const float* trow = a.t + (size_t)(lo + n) * At;
      float* urow = a.d_u + (size_t)(lo + n) * At;
      if (FAST) {
        float tv[MAX_HEADS][2];
#pragma unroll
        for (int k = 0; k < MAX_HEADS; ++k)
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const int j = lane + 64 * i;
            tv[k][i] = (k < NH && j < A) ? trow[k * A + j] : 0.f;
          }
#pragma unroll
        for (int k = 0; k < MAX_HEADS; ++k)
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const int j = lane + 64 * i;
            if (k < NH && j < A) {
              const float uv = ds[k] * w3s[k * A + j] * (1.f - tv[k][i] * tv[k][i]);
              urow[k * A + j] = uv;
              su[k][i] += uv;
              sw[k][i] += ds[k] * tv[k][i];
            }
          }
#pragma unroll
        for (int k = 0; k < MAX_HEADS; ++k) ssum[k] += ds[k];
      } else {
#pragma unroll
        for (int k = 0; k < MAX_HEADS; ++k) if (k < NH) {
          for (int j = lane; j < A; j += 64) {
            const float tv = trow[k * A + j];
            urow[k * A + j] = ds[k] * w3s[k * A + j] * (1.f - tv * tv);
          }
        }
      }
    }
    if (a.W4 && a.d_P && lane < C) {
      float v = 0.f;
#pragma unroll
      for (int c = 0; c < MAX_C; ++c) if (lane == c) v = an[0] * dL[c];
      a.d_P[(size_t)(lo + n) * C + lane] = v;
    }
    if (a.d_h) {
      float asum = 0.f;
#pragma unroll
      for (int k = 0; k < MAX_HEADS; ++k) if (k < NH) asum += an[k];
      asum *= inv_heads;
      float* drow = a.d_h + (size_t)(lo + n) * H;
      for (int j = lane; j < H; j += 64) {
        float v = asum * dzs[j];
        if (a.W4 && a.d_bag_logits) {
#pragma unroll
          for (int c = 0; c < MAX_C; ++c) if (c < C) v += W4s[c * H + j] * (an[0] * dL[c]);
        }
        drow[j] = a.accumulate_dh ? drow[j] + v : v;
      }
    }
  }
  if (FAST && a.psum) {                                      // block-uniform
    const int PW = 2 * At + NH;
    float* wred = red + NWAVE_B * NH + wave * PW;
#pragma unroll
    for (int k = 0; k < MAX_HEADS; ++k)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int j = lane + 64 * i;
        if (k < NH && j < A) { wred[k * A + j] = su[k][i]; wred[At + k * A + j] = sw[k][i]; }
      }
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < MAX_HEADS; ++k) if (k < NH) wred[2 * At + k] = ssum[k];
    }
    __syncthreads();
    const float* w0 = red + NWAVE_B * NH;
    for (int c = tid; c < PW; c += NTHR_B) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < NWAVE_B; ++w) v += w0[w * PW + c];   // fixed order: the bag's sum does not depend on timing
      a.psum[(size_t)b * PW + c] = v;
    }
  }
}

template <typename K>
int ensure_lds(K kernel, size_t bytes) {
  if (bytes > 160 * 1024) return ISIC_ERR_UNSUPPORTED;
  if (bytes > 48 * 1024) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)bytes) != hipSuccess)
      return ISIC_ERR_LAUNCH;
  }
  return ISIC_OK;
}

}  // namespace

extern "C" {

int isic_attn_pool_fwd(const float* h, const float* t, const float* w3, const float* b3, const float* W4,
                       const float* b4, const int64_t* offsets, int B, int H, int A, int heads, int C, int max_bag,
                       float* att, float* z, float* patch_logits, float* patch_probs, float* bag_logits,
                       float* bag_probs, void* stream) {
  ISIC_CHECK_ARG(B >= 0 && H > 0 && A > 0 && heads > 0 && max_bag >= 0);
  if (B == 0) return ISIC_OK;
  ISIC_CHECK_ARG(h && t && w3 && b3 && offsets && att);
  if (W4) { ISIC_CHECK_ARG(heads == 1 && C > 0 && b4); if (C > MAX_C) return ISIC_ERR_UNSUPPORTED; }
  if (H > 1024) return ISIC_ERR_UNSUPPORTED;
  for (int head0 = 0; head0 < heads; head0 += MAX_HEADS) {
    const int nh = heads - head0 < MAX_HEADS ? heads - head0 : MAX_HEADS;
    PoolArgs a;
    a.h = h; a.t = t; a.w3 = w3; a.b3 = b3; a.W4 = W4; a.b4 = b4; a.offsets = offsets;
    a.B = B; a.H = H; a.A = A; a.heads = nh; a.C = W4 ? C : 0; a.max_bag = max_bag;
    a.head0 = head0; a.heads_total = heads;
    a.att = att; a.z = z; a.patch_logits = patch_logits; a.patch_probs = patch_probs;
    a.bag_logits = bag_logits; a.bag_probs = bag_probs; a.z_accumulate = head0 > 0;
    const size_t lds = sizeof(float) * ((size_t)nh * A + (W4 ? (size_t)C * H : 0) + (size_t)max_bag * nh +
                                        2 * NWAVE * nh + NWAVE * MAX_C + (size_t)NWAVE * nh * H + 2 * MAX_HEADS +
                                        MAX_HEADS * NWAVE);
    int rc;
    if (!W4 && z && H <= 128 && A <= 128 && heads <= MAX_HEADS && !patch_logits && !patch_probs && !bag_logits && !bag_probs) {
      // GraphMIL form: the two-pass kernel
      const size_t lds2 = sizeof(float) * ((size_t)nh * A + (size_t)max_bag * nh + max_bag + 4 * (size_t)H + 2 * MAX_HEADS);
      rc = ensure_lds(attn_pool_fwd2_kernel, lds2);
      if (rc != ISIC_OK) return rc;
      hipLaunchKernelGGL(attn_pool_fwd2_kernel, dim3(B), dim3(NTHR), lds2, as_stream(stream), a);
      return isic_launch_status();
    }
#define LAUNCH_POOL(JH, JA)                                                                         \
  rc = ensure_lds(attn_pool_fwd_kernel<JH, JA>, lds);                                               \
  if (rc != ISIC_OK) return rc;                                                                     \
  hipLaunchKernelGGL((attn_pool_fwd_kernel<JH, JA>), dim3(B), dim3(NTHR), lds, as_stream(stream), a)
    if (H <= 128 && A <= 128) { LAUNCH_POOL(2, 2); }
    else if (H <= 128) { LAUNCH_POOL(2, 0); }
    else if (H <= 256) { LAUNCH_POOL(4, 0); }
    else if (H <= 512) { LAUNCH_POOL(8, 0); }
    else { LAUNCH_POOL(16, 0); }
#undef LAUNCH_POOL
    rc = isic_launch_status();
    if (rc != ISIC_OK) return rc;
  }
  return ISIC_OK;
}

int isic_attn_pool_bwd(const float* h, const float* t, const float* att, const float* patch_logits,
                       const float* w3, const float* W4, const int64_t* offsets, int B, int H, int A, int heads,
                       int C, int max_bag, const float* d_bag_logits, const float* d_z, float* d_h,
                       int accumulate_dh, float* d_u, float* d_s, float* d_P, void* stream) {
  return isic_attn_pool_bwd_sums(h, t, att, patch_logits, w3, W4, offsets, B, H, A, heads, C, max_bag, d_bag_logits, d_z, d_h,
                                 accumulate_dh, d_u, d_s, d_P, nullptr, stream);
}

int isic_attn_pool_bwd_sums(const float* h, const float* t, const float* att, const float* patch_logits,
                            const float* w3, const float* W4, const int64_t* offsets, int B, int H, int A, int heads,
                            int C, int max_bag, const float* d_bag_logits, const float* d_z, float* d_h,
                            int accumulate_dh, float* d_u, float* d_s, float* d_P, float* param_sums, void* stream) {
  ISIC_CHECK_ARG(B >= 0 && H > 0 && A > 0 && heads > 0 && max_bag >= 0);
  if (B == 0) return ISIC_OK;
  ISIC_CHECK_ARG(h && t && att && w3 && offsets);
  if (heads > MAX_HEADS) return ISIC_ERR_UNSUPPORTED;
  if (W4) { ISIC_CHECK_ARG(heads == 1 && C > 0 && patch_logits); if (C > MAX_C) return ISIC_ERR_UNSUPPORTED; }
  PoolBwdArgs a;
  a.h = h; a.t = t; a.att = att; a.P = patch_logits; a.w3 = w3; a.W4 = W4; a.offsets = offsets;
  a.B = B; a.H = H; a.A = A; a.heads = heads; a.C = W4 ? C : 0; a.max_bag = max_bag;
  a.d_bag_logits = d_bag_logits; a.d_z = d_z; a.d_h = d_h; a.accumulate_dh = accumulate_dh;
  a.d_u = d_u; a.d_s = d_s; a.d_P = d_P; a.psum = param_sums;
  if (param_sums && !(H <= 128 && A <= 128 && d_u)) return ISIC_ERR_UNSUPPORTED;      // the sums are taken by the two-columns-per-lane form
  const size_t lds = sizeof(float) * ((size_t)heads * A + (W4 ? (size_t)C * H : 0) + H + (size_t)max_bag * heads +
                                      NWAVE_B * heads + (param_sums ? (size_t)NWAVE_B * (2 * heads * A + heads) : 0));
  int rc;
  if (H <= 128 && A <= 128) {
    rc = ensure_lds(attn_pool_bwd_kernel<true>, lds);
    if (rc != ISIC_OK) return rc;
    hipLaunchKernelGGL(attn_pool_bwd_kernel<true>, dim3(B), dim3(NTHR_B), lds, as_stream(stream), a);
  } else {
    rc = ensure_lds(attn_pool_bwd_kernel<false>, lds);
    if (rc != ISIC_OK) return rc;
    hipLaunchKernelGGL(attn_pool_bwd_kernel<false>, dim3(B), dim3(NTHR_B), lds, as_stream(stream), a);
  }
  return isic_launch_status();
}

}  // extern "C"
