// preprocess_bwd.hip -- K8/K9/K10: per-Gaussian backward + camera-pose Jacobians.
//
// One thread per Gaussian:
//   1. sums the Gaussian's consecutive 48-byte partial records written by blend_bwd (no atomics anywhere);
//   2. moments -> dL/d{mean2D, conic, opacity, rgb};
//   3. conic -> cov2D -> (J, Wc, Sigma) -> dL/d{means3D, scales, rotations | cov3D_precomp};
//   4. pixel centre -> p_hom -> dL/dmeans3D; colour -> dL/d{shs | colors_precomp}, view direction -> dL/dmeans3D;
//   5. the camera side of every one of those products: dL/d{viewmatrix, projmatrix, intrinsic, campos,
//      shift_factors} (35 non-zero scalars) reduced wave -> workgroup -> one slab row per workgroup; a second
//      tiny kernel adds the rows in fp64, so pose gradients are deterministic and do not lose bits to a
//      half-million-term fp32 chain.
// The forward quantities are recomputed from the inputs (cheaper than 100+ B/Gaussian of saved state).
#include "bags_common.h"
#include "sh_basis.h"

struct CamConstB {
    float v[16], m[16], k[16];
    float campos[3];
    float sf[3];
};

// sum over the 64 lanes on DPP (row_shr 1/2/4/8 inside the 16-lane rows, row_bcast:15 / row_bcast:31 across them: six
// v_add_f32_dpp); the total is valid in lane 63.  Every lane must be active.
__device__ __forceinline__ float wave_total_f(float x)
{
#define DPP_ADD(ctrl, rmask) x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), ctrl, rmask, 0xf, false))
    DPP_ADD(0x111, 0xf); DPP_ADD(0x112, 0xf); DPP_ADD(0x114, 0xf); DPP_ADD(0x118, 0xf); DPP_ADD(0x142, 0xa); DPP_ADD(0x143, 0xc);
#undef DPP_ADD
    return x;
}
__device__ __forceinline__ float wave_sum(float x)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) x += __shfl_xor(x, d);
    return x;
}

// K8a: each Gaussian's consecutive partial records (written by blend_bwd at its emission slots) summed in list order.
// Light kernel (high occupancy, two records in flight per lane): the record stream is the only traffic.  Gaussians with
// more than SUM_COOP records are summed by the whole wave (lane k takes records k, k+64, ...; fixed-order shuffle
// tree), so a splat covering thousands of tiles does not serialise one lane.
#define SUM_COOP 64
#define RQ (PART_FLOATS / 4)          // float4s per record
// The summation itself (whole wave: every lane calls it, lanes without a Gaussian with nrec = 0).
// LIVE (dense-scene mode, BagsBackwardArgs.dense_per_tile): live[e] != 0 iff blend_bwd wrote record e; the others -- instances behind
// their tile's deepest contributor, 85 % of them at 1800 instances per tile -- hold whatever the buffer held before and are not
// read.  The live ones are added in the same (slot) order as without the map, so the sums are the same bit for bit: the records
// skipped are the ones that were zeros.
template <bool LIVE>
__device__ __forceinline__ void sum_records(u32 nrec, u32 first, const float* __restrict__ partials, const unsigned char* __restrict__ live,
                                            float4& s0, float4& s1, float4& s2)
{
    const int lane = threadIdx.x & 63;
    s0 = make_float4(0.f, 0.f, 0.f, 0.f); s1 = s0; s2 = s0;
    if (LIVE && nrec > 0 && nrec <= SUM_COOP) {
        // the Gaussian's <= 64 marks first (consecutive bytes, all requested before the first is looked at), then its live records two
        // at a time as below
        // 64 bytes from the Gaussian's first mark on, as four 16-byte loads at whatever alignment, all in flight at once (bytes past its last
        // mark belong to the next Gaussians or to the 256 bytes of slack behind the map, bags_backward_workspace_size).  Until the end of round 5
        // this was a loop of eight byte loads per trip: four dependent round trips for the 30 records of a Gaussian at sm 2.0.
        // A mark is 0 or 1: the four of a word gather into four bits with one multiplication.
        struct __attribute__((packed, aligned(1))) U4 { u32 x, y, z, w; };
        const U4* lp = reinterpret_cast<const U4*>(live + first);
        const U4 c0 = lp[0], c1 = lp[1], c2 = lp[2], c3 = lp[3];
        // (`& 0x01010101` first: the bytes behind the map's last mark are not marks -- uninitialised slack -- and a byte above 1 would carry into
        // the bits of its word's real marks; found by tools/fuzz_paths.py --cross-dense as one Gaussian in a few thousand, differently every run)
        auto nib = [](u32 w) -> u64 { return (u64)((((w & 0x01010101u) * 0x00204081u) >> 21) & 0xFu); };
        u64 lm = nib(c0.x) | (nib(c0.y) << 4) | (nib(c0.z) << 8) | (nib(c0.w) << 12) | (nib(c1.x) << 16) | (nib(c1.y) << 20) | (nib(c1.z) << 24) |
                 (nib(c1.w) << 28) | (nib(c2.x) << 32) | (nib(c2.y) << 36) | (nib(c2.z) << 40) | (nib(c2.w) << 44) | (nib(c3.x) << 48) |
                 (nib(c3.y) << 52) | (nib(c3.z) << 56) | (nib(c3.w) << 60);
        lm &= (nrec >= 64u) ? ~0ull : ((1ull << nrec) - 1ull);
        const float4* rec = reinterpret_cast<const float4*>(partials + (size_t)first * PART_FLOATS);
        while (lm) {
            const u32 r = (u32)__builtin_ctzll(lm); lm &= lm - 1ull;
            const bool two = lm != 0ull;
            const u32 r1 = two ? (u32)__builtin_ctzll(lm) : r; if (two) lm &= lm - 1ull;
            const float4 a0 = rec[RQ * r], b0 = rec[RQ * r + 1], c0 = rec[RQ * r + 2];
            const float4 a1 = rec[RQ * r1], b1 = rec[RQ * r1 + 1], c1 = rec[RQ * r1 + 2];
            // (both records requested before either is added: left alone the compiler sinks the second one's loads into `if (two)`, behind
            // the first one's wait)
            asm volatile("" :: "v"(a1.x), "v"(b1.x), "v"(c1.x));
            s0.x += a0.x; s0.y += a0.y; s0.z += a0.z; s0.w += a0.w;
            s1.x += b0.x; s1.y += b0.y; s1.z += b0.z; s1.w += b0.w;
            s2.x += c0.x; s2.y += c0.y; s2.z += c0.z;
            if (two) {
                s0.x += a1.x; s0.y += a1.y; s0.z += a1.z; s0.w += a1.w;
                s1.x += b1.x; s1.y += b1.y; s1.z += b1.z; s1.w += b1.w;
                s2.x += c1.x; s2.y += c1.y; s2.z += c1.z;
            }
        }
    }
    if (!LIVE && nrec > 0 && nrec <= SUM_COOP) {
        const float4* rec = reinterpret_cast<const float4*>(partials + (size_t)first * PART_FLOATS);
        // (four records per trip with the missing ones read again and dropped: 0.0608 -> 0.0635 ms, profiles/r05/ab_k1.txt)
        u32 r = 0;
        for (; r + 1 < nrec; r += 2) {
            const float4 a0 = rec[RQ * r], b0 = rec[RQ * r + 1], c0 = rec[RQ * r + 2];
            const float4 a1 = rec[RQ * r + RQ], b1 = rec[RQ * r + RQ + 1], c1 = rec[RQ * r + RQ + 2];
            s0.x += a0.x; s0.y += a0.y; s0.z += a0.z; s0.w += a0.w;
            s1.x += b0.x; s1.y += b0.y; s1.z += b0.z; s1.w += b0.w;
            s2.x += c0.x; s2.y += c0.y; s2.z += c0.z;
            s0.x += a1.x; s0.y += a1.y; s0.z += a1.z; s0.w += a1.w;
            s1.x += b1.x; s1.y += b1.y; s1.z += b1.z; s1.w += b1.w;
            s2.x += c1.x; s2.y += c1.y; s2.z += c1.z;
        }
        if (r < nrec) {
            const float4 a0 = rec[RQ * r], b0 = rec[RQ * r + 1], c0 = rec[RQ * r + 2];
            s0.x += a0.x; s0.y += a0.y; s0.z += a0.z; s0.w += a0.w;
            s1.x += b0.x; s1.y += b0.y; s1.z += b0.z; s1.w += b0.w;
            s2.x += c0.x; s2.y += c0.y; s2.z += c0.z;
        }
    }
    u64 big = __ballot(nrec > SUM_COOP);
    while (big) {
        const int src = __ffsll((long long)big) - 1;
        big &= big - 1;
        const u32 bn = __shfl(nrec, src), bf = __shfl(first, src);
        const float4* rec = reinterpret_cast<const float4*>(partials + (size_t)bf * PART_FLOATS);
        float v[11];
#pragma unroll
        for (int t = 0; t < 11; ++t) v[t] = 0.f;
        for (u32 r = lane; r < bn; r += 64) {
            if (LIVE && live[bf + r] == 0) continue;
            const float4 a0 = rec[RQ * r], b0 = rec[RQ * r + 1], c0 = rec[RQ * r + 2];
            v[0] += a0.x; v[1] += a0.y; v[2] += a0.z; v[3] += a0.w; v[4] += b0.x; v[5] += b0.y; v[6] += b0.z; v[7] += b0.w;
            v[8] += c0.x; v[9] += c0.y; v[10] += c0.z;
        }
#pragma unroll
        for (int t = 0; t < 11; ++t) {
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) v[t] += __shfl_xor(v[t], d);
        }
        if (lane == src) {
            s0 = make_float4(v[0], v[1], v[2], v[3]); s1 = make_float4(v[4], v[5], v[6], v[7]);
            s2 = make_float4(v[8], v[9], v[10], 0.f);
        }
    }
}
#define PRE_BWD_WAVES 4        // 128 VGPRs without spills and 4 workgroups per CU (58 us; left free -- 146 VGPRs, 3 per CU -- 76 us, 5 per CU 105 us).
                               // The kernel sums the records itself: as two kernels (sum_partials + this one, the sums through memory) 76 us;
                               // folding the pose reduction in as well was correct and no faster (profiles/r04/ab_pose_fold.txt)

// Slab column t (summed over all Gaussians) -> its place in the five pose tensors.
// slab layout: [0..11] viewmatrix rows 0..3 x cols 0..2, [12..23] projmatrix rows 0..3 x cols 0,1,3,
//              [24] k0 [25] k5 [26] k8 [27] k9 [28] k11, [29..31] campos, [32..34] shift_factors.
// The entries of the 4x4 outputs no Gaussian contributes to are written as zeros by the column next to them.
__device__ __forceinline__ void pose_write_out(const int t, const float val, float* __restrict__ g_view, float* __restrict__ g_proj,
                                               float* __restrict__ g_intr, float* __restrict__ g_campos, float* __restrict__ g_shift)
{
    if (t < 12) {
        if (g_view) { g_view[(t / 3) * 4 + t % 3] = val; if (t % 3 == 2) g_view[(t / 3) * 4 + 3] = 0.f; }
    } else if (t < 24) {
        const int u = t - 12, r = u / 3, k = u % 3;
        if (g_proj) { g_proj[r * 4 + (k == 2 ? 3 : k)] = val; if (k == 2) g_proj[r * 4 + 2] = 0.f; }
    } else if (t < 29) {
        const int at[5] = {0, 5, 8, 9, 11};
        if (g_intr) {
            g_intr[at[t - 24]] = val;
            if (t == 24) {
#pragma unroll
                for (int i = 0; i < 16; ++i) if (i != 0 && i != 5 && i != 8 && i != 9 && i != 11) g_intr[i] = 0.f;
            }
        }
    } else if (t < 32) {
        if (g_campos) g_campos[t - 29] = val;
    } else if (t < 35) {
        if (g_shift) g_shift[t - 32] = val;
    }
}

// ACCUM (BagsBackwardArgs.accumulate): the seven Gaussian-parameter gradients are ADDED to what their buffers hold (several
// views of one step accumulate in place: no separate add pass per view); means2D / densify / pose outputs are overwritten.
template <bool COV3D, bool ACCUM, bool LIVE>   // COV3D: precomputed 3D covariances instead of scales + rotations (uniform: no branch at the top)
__global__ void __launch_bounds__(256, PRE_BWD_WAVES)
preprocess_bwd_kernel(int P, int M, int deg, int W, int H, float tanfovx, float tanfovy, float mod, int clamp_stock, int conic_stock,
                      const float* __restrict__ means3D, const float* __restrict__ shift_factors,
                      const float* __restrict__ shs, const float* __restrict__ colors_precomp,
                      const float* __restrict__ scales, const float* __restrict__ rotations,
                      const float* __restrict__ cov3D_precomp, const float* __restrict__ viewmatrix,
                      const float* __restrict__ projmatrix, const float* __restrict__ intrinsic,
                      const float* __restrict__ campos_p,
                      const float* __restrict__ opacities,
                      const u32* __restrict__ tiles_touched, const u32* __restrict__ inst_off, const u32* __restrict__ local_off,
                      const u32* __restrict__ block_base, int per_block, const float* __restrict__ shjac,
                      const float* __restrict__ partials, const unsigned char* __restrict__ live_map, float* __restrict__ pose_slab,
                      float* __restrict__ g_means3D, float* __restrict__ g_means2D, float* __restrict__ g_densify,
                      float* __restrict__ g_shs, float* __restrict__ g_shs_rest, float* __restrict__ g_dldc, float* __restrict__ g_colors,
                      float* __restrict__ g_opac,
                      float* __restrict__ g_scales, float* __restrict__ g_rot, float* __restrict__ g_cov3D)
{
    // A wave's life in this kernel is a handful of memory round trips, not arithmetic (55 % of the wave cycles were spent
    // waiting): round 2 had SEVEN of them in series at the top -- three for the camera constants (vector loads -> LDS ->
    // barrier), one for the SH rows (12 loads -> wait -> ds_write), one for the word that says whether the Gaussian is
    // visible, two for its inputs.  Now everything is requested at once: the camera constants are read with wave-uniform
    // indices from the kernel's own pointers (scalar loads, no LDS, no barrier), the SH rows go to LDS by LDS-DMA (no
    // registers, no wait here), the visibility word and every input row are requested unconditionally (clamped index; a
    // culled Gaussian's 104 bytes are read for nothing), and there is ONE wait in front of the arithmetic.
    __shared__ float wpose[4][48];                      // [0..34] the slab row, [35..43] second contribution to [0..8]
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const float* v = viewmatrix; const float* m = projmatrix; const float* k = intrinsic;
    const float sf0 = shift_factors ? shift_factors[0] : 0.0f, sf1 = shift_factors ? shift_factors[1] : 0.0f,
                sf2 = shift_factors ? shift_factors[2] : 0.0f;
    const float cpx = campos_p[0], cpy = campos_p[1], cpz = campos_p[2];
    // The SH-GRADIENT rows of the workgroup's 256 Gaussians are ONE contiguous 48 KB span of dL/dshs.  A thread writing its
    // own 192-byte row 16 bytes at a time puts 64 separate requests per instruction on the L2 channels (12 such instructions
    // per Gaussian), so the span leaves as whole lines: thread t stores float4 number k * 256 + t of it.  A gradient row is an
    // outer product, dL/dsh[t][c] = basis_t x dL/dcolour_c, so what the storing thread needs of ANOTHER thread's Gaussian is
    // 16 + 3 floats, not 48: 20 KB of LDS per workgroup instead of 48 (round 2 staged the finished rows; and up to round 2
    // the SH rows themselves came IN through the same 48 KB -- since round 3 the forward leaves d(colour)/d(direction) behind,
    // 48 bytes per Gaussian, and the backward does not read the 192-byte row a second time).  With 108 VGPRs that makes four
    // workgroups per CU (three before, both by registers and by LDS).
    __shared__ float srow[256][20];                     // basis[16] (zero beyond the active degree), dL/dcolour[3], pad
    const bool stage = (M == 16) && (colors_precomp == nullptr) && (shs != nullptr) && (g_shs != nullptr);      // uniform
    const size_t base4 = (size_t)blockIdx.x * (256 * 12), lim4 = (size_t)P * 12;

    float dmx = 0.f, dmy = 0.f, dmz = 0.f;           // dL/dmeans3D
    float gm2x = 0.f, gm2y = 0.f, gdx = 0.f, gdy = 0.f, gop = 0.f;
    float gs0 = 0.f, gs1 = 0.f, gs2 = 0.f, gqr = 0.f, gqx = 0.f, gqy = 0.f, gqz = 0.f;
    float gc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float drgb[3] = {0.f, 0.f, 0.f};
    // Every input row of the Gaussian in one batch, before anything is looked at (index clamped: P > 0 here)
    const size_t ic = (size_t)(i < P ? i : P - 1);
    // (nothing is read from the Gaussian's 64-byte geometry line any more: the conic is re-derived below bit for bit, the
    // visibility comes from the compact tiles_touched array, the SH clamp bits ride in the tenth word of shjac)
    const u32 n_inst = tiles_touched[ic];
    // first record of the Gaussian: radix path inst_off[i]; tile-binned path block_base[block] + local_off[i], the block being
    // uniform over the workgroup (per_block is a multiple of 1024)
    u32 first_rec = inst_off ? inst_off[ic] : local_off[ic];
    if (!inst_off) first_rec += block_base[(blockIdx.x * 256u) / (u32)per_block];
    const float opac = opacities[ic];
    float x = means3D[3 * ic + 0], y = means3D[3 * ic + 1], z = means3D[3 * ic + 2];
    float in_s0 = 0.f, in_s1 = 0.f, in_s2 = 0.f;
    float4 in_q = make_float4(0.f, 0.f, 0.f, 0.f);
    float in_c[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (COV3D) {
        const float* c = cov3D_precomp + 6 * ic;
#pragma unroll
        for (int t = 0; t < 6; ++t) in_c[t] = c[t];
    } else {
        in_s0 = scales[3 * ic + 0]; in_s1 = scales[3 * ic + 1]; in_s2 = scales[3 * ic + 2];
        in_q = reinterpret_cast<const float4*>(rotations)[ic];
    }
    float mj[10];                                                                           // K1's d(colour)/d(direction) + clamp bits (SH path)
#pragma unroll
    for (int t = 0; t < 10; ++t) mj[t] = shjac[10 * ic + t];
    // the Gaussian's records (blend_bwd's, `partials` is the record array) summed here, with every other input in flight
    float4 sm_a, sm_b, sm_c;
    __builtin_amdgcn_sched_barrier(0);
    sum_records<LIVE>(i < P ? n_inst : 0u, first_rec, partials, live_map, sm_a, sm_b, sm_c);
    // every one of those loads is IN FLIGHT before the first of them is waited for (the compiler otherwise sinks the ones only
    // the visible branch needs behind the visibility test: one more round trip per group)
    asm volatile("" :: "v"(opac), "v"(n_inst), "v"(mj[9]), "v"(x), "v"(y), "v"(z), "v"(in_s0), "v"(in_s1), "v"(in_s2), "v"(in_q.x), "v"(in_c[0]),
                 "v"(in_c[3]), "v"(sm_a.x), "v"(sm_b.x), "v"(sm_c.x), "v"(mj[0]), "v"(mj[3]), "v"(mj[6]));
    const bool live = (i < P) && (n_inst > 0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // Pose Jacobians (35 scalars per Gaussian) leave the registers the moment they exist: wave total on DPP (six v_add_f32_dpp,
    // total in lane 63) -> wpose.  Round 3 formed all 35 inside `if (live)` and reduced them behind it: 35 values live across
    // the join on top of everything else put the kernel at 140+ registers against the 128 that four workgroups per CU allow
    // (12-18 spilled: 50-70 MB of scratch traffic per launch).  For that the geometry chain runs in EVERY lane (a culled
    // Gaussian's lane computes on its real inputs; whatever comes out -- Inf, NaN -- is replaced by 0 in the select below
    // and never stored), so the DPP rows are complete.  Slots 35..43 take the second contribution to viewmatrix[0..2][0..2].
    auto red = [&](const int t, const float val) {
        const float tot = wave_total_f(live ? val : 0.f);
        if (lane == 63) wpose[wave][t] = tot;
    };
    {
        // ---- 1. the per-Gaussian sums of the per-instance records
        float s[12];
        {
            const float4 a = sm_a, b = sm_b, c = sm_c;
            s[0] = a.x; s[1] = a.y; s[2] = a.z; s[3] = a.w; s[4] = b.x; s[5] = b.y; s[6] = b.z; s[7] = b.w;
            s[8] = c.x; s[9] = c.y; s[10] = c.z; s[11] = 0.f;
        }
        drgb[0] = s[0]; drgb[1] = s[1]; drgb[2] = s[2];
        // the records carry sum q = o * sum G dL/dalpha (blend_bwd forms q from the unclamped o G): one division per Gaussian
        // here instead of one per record there; a Gaussian with a contributing pixel has o >= 1/255
        gop = (opac > 0.f) ? s[3] / opac : 0.f;
        const float Mx = s[4], My = s[5], Mxx = s[6], Mxy = s[7], Myy = s[8];
        // ---- 2. moments -> screen-space gradients (the centre's, which needs the conic, follows the covariance below)
        const float gA = -0.5f * Mxx, gB = -Mxy, gC = -0.5f * Myy;   // dL/dconic
        // (the abs sums were taken on the conic pre-scaled by log2 e)
        gdx = s[9] * (0.6931471805599453f * 0.5f * (float)W); gdy = s[10] * (0.6931471805599453f * 0.5f * (float)H);

        // ---- recompute the forward chain
        const float tx = x * v[0] + y * v[4] + z * v[8] + v[12];
        const float ty = x * v[1] + y * v[5] + z * v[9] + v[13];
        const float tz = x * v[2] + y * v[6] + z * v[10] + v[14];
        const float rho = sqrtf(tx * tx + ty * ty + 1e-20f);
        const float theta = det_atan2_pos(rho, tz);
        const float th2 = theta * theta, th3 = th2 * theta;
        const float shift = sf0 * th3 + sf1 * (th3 * th2) + sf2 * (th3 * th2 * th2);
        const float tzs = tz + shift;
        const float hx = x * m[0] + y * m[4] + z * m[8] + m[12] + shift * k[8];
        const float hy = x * m[1] + y * m[5] + z * m[9] + m[13] + shift * k[9];
        const float hw = x * m[3] + y * m[7] + z * m[11] + m[15] + shift * k[11];
        const float pw = 1.0f / (hw + 1e-7f);

        float c0, c1, c2, c3, c4, c5;
        float s0 = 0, s1 = 0, s2 = 0, qr = 0, qx = 0, qy = 0, qz = 0;
        float r00 = 0, r01 = 0, r02 = 0, r10 = 0, r11 = 0, r12 = 0, r20 = 0, r21 = 0, r22 = 0;
        if (COV3D) {
            c0 = in_c[0]; c1 = in_c[1]; c2 = in_c[2]; c3 = in_c[3]; c4 = in_c[4]; c5 = in_c[5];
        } else {
            s0 = in_s0 * mod; s1 = in_s1 * mod; s2 = in_s2 * mod;
            const float4 q = in_q;
            qr = q.x; qx = q.y; qy = q.z; qz = q.w;
            r00 = 1.0f - 2.0f * (qy * qy + qz * qz); r01 = 2.0f * (qx * qy - qr * qz); r02 = 2.0f * (qx * qz + qr * qy);
            r10 = 2.0f * (qx * qy + qr * qz); r11 = 1.0f - 2.0f * (qx * qx + qz * qz); r12 = 2.0f * (qy * qz - qr * qx);
            r20 = 2.0f * (qx * qz - qr * qy); r21 = 2.0f * (qy * qz + qr * qx); r22 = 1.0f - 2.0f * (qx * qx + qy * qy);
            const float l00 = r00 * s0, l01 = r01 * s1, l02 = r02 * s2;
            const float l10 = r10 * s0, l11 = r11 * s1, l12 = r12 * s2;
            const float l20 = r20 * s0, l21 = r21 * s1, l22 = r22 * s2;
            c0 = l00 * l00 + l01 * l01 + l02 * l02; c1 = l00 * l10 + l01 * l11 + l02 * l12;
            c2 = l00 * l20 + l01 * l21 + l02 * l22; c3 = l10 * l10 + l11 * l11 + l12 * l12;
            c4 = l10 * l20 + l11 * l21 + l12 * l22; c5 = l20 * l20 + l21 * l21 + l22 * l22;
        }
        const float fx = k[0] * (0.5f * (float)W), fy = k[5] * (0.5f * (float)H);
        const float limx = 1.3f * tanfovx, limy = 1.3f * tanfovy;
        const float itz = 1.0f / tzs;
        const float txtz = tx / tzs, tytz = ty / tzs;                 // same expressions as preprocess_fwd
        const bool clx = (txtz < -limx) || (txtz > limx), cly = (tytz < -limy) || (tytz > limy);
        const float ux = fminf(limx, fmaxf(-limx, txtz)), uy = fminf(limy, fmaxf(-limy, tytz));
        const float itz2 = itz * itz;
        const float j00 = fx * itz, j02 = -(fx * (ux * tzs)) * itz2, j11 = fy * itz, j12 = -(fy * (uy * tzs)) * itz2;
        const float a00 = j00 * v[0] + j02 * v[2], a01 = j00 * v[4] + j02 * v[6], a02 = j00 * v[8] + j02 * v[10];
        const float a10 = j11 * v[1] + j12 * v[2], a11 = j11 * v[5] + j12 * v[6], a12 = j11 * v[9] + j12 * v[10];
        const float b00 = a00 * c0 + a01 * c1 + a02 * c2, b01 = a00 * c1 + a01 * c3 + a02 * c4, b02 = a00 * c2 + a01 * c4 + a02 * c5;
        const float b10 = a10 * c0 + a11 * c1 + a12 * c2, b11 = a10 * c1 + a11 * c3 + a12 * c4, b12 = a10 * c2 + a11 * c4 + a12 * c5;
        const float cxx = b00 * a00 + b01 * a01 + b02 * a02 + 0.3f;
        const float cxy = b00 * a10 + b01 * a11 + b02 * a12;
        const float cyy = b10 * a10 + b11 * a11 + b12 * a12 + 0.3f;
        // ---- 3. conic -> cov2D.  For needle-shaped splats (lambda1 >> lambda2) dL/dcov2D is ~ g n n^T (n = thin axis) and
        // the next step multiplies it by B = A Sigma, whose component along the long axis is lambda1/lambda2 times larger
        // than the one that matters: the long-axis component of dL/dcov2D must be accurate to ~1e-7 of the WHOLE matrix.
        // The expanded polynomial -Q G Q (three products of size |G|/lambda2^2 per entry) is not, and gave O(1) errors at
        // 1500:1 anisotropy; the two-step form below (trace term first, then one two-term difference per entry) keeps
        // that component at rounding level.  It is the order reverse-mode differentiation of the forward lines produces.
        // The file is compiled with -ffp-contract=off so cov2D and det are bit-identical to what preprocess_fwd used.
        const float det = cxx * cyy - cxy * cxy;
        const float di = 1.0f / det;
        // BagsSettings.conic_grad (decision D9).  Upstream's computeCov2DCUDA backward multiplies every term by denom2inv =
        // 1 / (det^2 + 1e-7) where the derivative of the inverse has 1 / det^2 -- in the two-step form below that is 1 / det ->
        // det * denom2inv and 1 / det^2 -> denom2inv (the same three sums as upstream's expanded polynomial, term for term).
        // BAGS_CONIC_GRAD_EXACT: the exact derivative.  The forward's conic keeps the exact 1 / det either way.
        const float di2 = conic_stock ? 1.0f / (det * det + 1.0e-7f) : di * di;
        const float dig = conic_stock ? det * di2 : di;
        {   // K1's conic = (cyy, -cxy, cxx) / det from the same operations (contraction off in both files): bit-identical
            const float con_a = cyy * di, con_b = -cxy * di, con_c = cxx * di;
            const float dpx = -(con_a * Mx + con_b * My);     // dL/d centre (pixel units)
            const float dpy = -(con_c * My + con_b * Mx);
            gm2x = dpx * (0.5f * (float)W); gm2y = dpy * (0.5f * (float)H);
        }
        const float ddet = -((gA * cyy - gB * cxy + gC * cxx) * di2);
        const float dcxx = gC * dig + ddet * cyy;
        const float dcyy = gA * dig + ddet * cxx;
        const float dcxy = -(gB * dig) - 2.f * (ddet * cxy);
        // cov2D -> Sigma (unique entries)
        gc[0] = dcxx * a00 * a00 + dcxy * a00 * a10 + dcyy * a10 * a10;
        gc[3] = dcxx * a01 * a01 + dcxy * a01 * a11 + dcyy * a11 * a11;
        gc[5] = dcxx * a02 * a02 + dcxy * a02 * a12 + dcyy * a12 * a12;
        gc[1] = 2.f * dcxx * a00 * a01 + dcxy * (a00 * a11 + a01 * a10) + 2.f * dcyy * a10 * a11;
        gc[2] = 2.f * dcxx * a00 * a02 + dcxy * (a00 * a12 + a02 * a10) + 2.f * dcyy * a10 * a12;
        gc[4] = 2.f * dcxx * a01 * a02 + dcxy * (a01 * a12 + a02 * a11) + 2.f * dcyy * a11 * a12;
        // cov2D -> A = J Wc
        const float dA00 = 2.f * dcxx * b00 + dcxy * b10, dA01 = 2.f * dcxx * b01 + dcxy * b11, dA02 = 2.f * dcxx * b02 + dcxy * b12;
        const float dA10 = 2.f * dcyy * b10 + dcxy * b00, dA11 = 2.f * dcyy * b11 + dcxy * b01, dA12 = 2.f * dcyy * b12 + dcxy * b02;
        // A -> J, viewmatrix rotation block
        const float dj00 = dA00 * v[0] + dA01 * v[4] + dA02 * v[8];
        const float dj02 = dA00 * v[2] + dA01 * v[6] + dA02 * v[10];
        const float dj11 = dA10 * v[1] + dA11 * v[5] + dA12 * v[9];
        const float dj12 = dA10 * v[2] + dA11 * v[6] + dA12 * v[10];
        // pose slab: [0..11] viewmatrix (rows 0..3 x cols 0..2), [12..23] projmatrix (rows 0..3 x cols 0,1,3),
        //            [24] k0 [25] k5 [26] k8 [27] k9 [28] k11, [29..31] campos, [32..34] shift_factors;
        //            [35..43] second contribution to [0..8] (added when the slab row is written)
        red(0, dA00 * j00); red(1, dA10 * j11); red(2, dA00 * j02 + dA10 * j12);      // v[0], v[1], v[2]
        red(3, dA01 * j00); red(4, dA11 * j11); red(5, dA01 * j02 + dA11 * j12);      // v[4], v[5], v[6]
        red(6, dA02 * j00); red(7, dA12 * j11); red(8, dA02 * j02 + dA12 * j12);      // v[8], v[9], v[10]
        // J -> focal lengths, view-space point
        const float dfx = dj00 * itz - dj02 * ux * itz;
        const float dfy = dj11 * itz - dj12 * uy * itz;
        red(24, dfx * (0.5f * (float)W));
        red(25, dfy * (0.5f * (float)H));
        const float dux = -dj02 * fx * itz, duy = -dj12 * fy * itz;
        const float ditz = dj00 * fx - dj02 * fx * ux + dj11 * fy - dj12 * fy * uy;
        float dtx = 0.f, dty = 0.f, dtz = 0.f;
        float dtzs = -ditz * itz * itz;
        // Frustum clamp (BagsSettings.clamp_grad).  Upstream's computeCov2DCUDA backward, which the reference's fork inherits
        // (README.md:126): dL/dt.x = x_grad_mul * (-h_x / t.z^2) dL/dJ02 and dL/dt.z takes (2 h_x t.x / t.z^3) dL/dJ02 with the
        // CLAMPED t.x (= ux tzs here) as a constant -- i.e. the unclamped formula with t.x replaced.  BAGS_CLAMP_GRAD_EXACT drops
        // that term for a clamped axis (the clamped t.x is +-1.3 tanfov t.z: J02 = -fx ux / t.z, already in `ditz`).
        if (!clx) { dtx += dux * itz; dtzs -= dux * tx * itz * itz; }
        else if (clamp_stock) dtzs -= dux * (ux * tzs) * itz * itz;
        if (!cly) { dty += duy * itz; dtzs -= duy * ty * itz * itz; }
        else if (clamp_stock) dtzs -= duy * (uy * tzs) * itz * itz;

        // ---- 4. pixel centre -> homogeneous point
        const float dhx = gm2x * pw, dhy = gm2y * pw;
        const float dhw = -(gm2x * hx + gm2y * hy) * pw * pw;
        dmx = dhx * m[0] + dhy * m[1] + dhw * m[3];
        dmy = dhx * m[4] + dhy * m[5] + dhw * m[7];
        dmz = dhx * m[8] + dhy * m[9] + dhw * m[11];
        red(12, dhx * x); red(13, dhy * x); red(14, dhw * x);       // m[0], m[1], m[3]
        red(15, dhx * y); red(16, dhy * y); red(17, dhw * y);       // m[4], m[5], m[7]
        red(18, dhx * z); red(19, dhy * z); red(20, dhw * z);       // m[8], m[9], m[11]
        red(21, dhx);     red(22, dhy);     red(23, dhw);           // m[12], m[13], m[15]
        const float dshift = dhx * k[8] + dhy * k[9] + dhw * k[11] + dtzs;
        red(26, dhx * shift); red(27, dhy * shift); red(28, dhw * shift);
        dtz += dtzs;
        // shift polynomial in theta = atan2(rho, tz)
        red(32, dshift * th3); red(33, dshift * th3 * th2); red(34, dshift * th3 * th2 * th2);
        const float dtheta = dshift * (3.f * sf0 * th2 + 5.f * sf1 * th2 * th2 + 7.f * sf2 * th2 * th2 * th2);
        const float ir2 = 1.0f / (rho * rho + tz * tz);
        const float drho = dtheta * tz * ir2;
        dtz -= dtheta * rho * ir2;
        dtx += drho * tx / rho; dty += drho * ty / rho;
        // view-space point -> world point, viewmatrix
        dmx += dtx * v[0] + dty * v[1] + dtz * v[2];
        dmy += dtx * v[4] + dty * v[5] + dtz * v[6];
        dmz += dtx * v[8] + dty * v[9] + dtz * v[10];
        red(35, dtx * x); red(36, dty * x); red(37, dtz * x);
        red(38, dtx * y); red(39, dty * y); red(40, dtz * y);
        red(41, dtx * z); red(42, dty * z); red(43, dtz * z);
        red(9, dtx); red(10, dty); red(11, dtz);                    // v[12], v[13], v[14]

        // ---- Sigma -> scales, rotation
        if (!COV3D) {
            const float S00 = 2.f * gc[0], S01 = gc[1], S02 = gc[2], S11 = 2.f * gc[3], S12 = gc[4], S22 = 2.f * gc[5];
            const float l00 = r00 * s0, l01 = r01 * s1, l02 = r02 * s2;
            const float l10 = r10 * s0, l11 = r11 * s1, l12 = r12 * s2;
            const float l20 = r20 * s0, l21 = r21 * s1, l22 = r22 * s2;
            const float dl00 = S00 * l00 + S01 * l10 + S02 * l20, dl01 = S00 * l01 + S01 * l11 + S02 * l21, dl02 = S00 * l02 + S01 * l12 + S02 * l22;
            const float dl10 = S01 * l00 + S11 * l10 + S12 * l20, dl11 = S01 * l01 + S11 * l11 + S12 * l21, dl12 = S01 * l02 + S11 * l12 + S12 * l22;
            const float dl20 = S02 * l00 + S12 * l10 + S22 * l20, dl21 = S02 * l01 + S12 * l11 + S22 * l21, dl22 = S02 * l02 + S12 * l12 + S22 * l22;
            gs0 = mod * (dl00 * r00 + dl10 * r10 + dl20 * r20);
            gs1 = mod * (dl01 * r01 + dl11 * r11 + dl21 * r21);
            gs2 = mod * (dl02 * r02 + dl12 * r12 + dl22 * r22);
            const float d00 = dl00 * s0, d01 = dl01 * s1, d02 = dl02 * s2;
            const float d10 = dl10 * s0, d11 = dl11 * s1, d12 = dl12 * s2;
            const float d20 = dl20 * s0, d21 = dl21 * s1, d22 = dl22 * s2;
            gqr = 2.f * (-qz * d01 + qy * d02 + qz * d10 - qx * d12 - qy * d20 + qx * d21);
            gqx = 2.f * (qy * d01 + qz * d02 + qy * d10 - 2.f * qx * d11 - qr * d12 + qz * d20 + qr * d21 - 2.f * qx * d22);
            gqy = 2.f * (-2.f * qy * d00 + qx * d01 + qr * d02 + qx * d10 + qz * d12 - qr * d20 + qz * d21 - 2.f * qy * d22);
            gqz = 2.f * (-2.f * qz * d00 - qr * d01 + qx * d02 + qr * d10 - 2.f * qz * d11 + qy * d12 + qx * d20 + qy * d21);
        }
        // a culled Gaussian's lane went through all of the above on its real inputs: none of it is kept
        if (!live) {
            dmx = dmy = dmz = 0.f; gm2x = gm2y = gdx = gdy = gop = 0.f; gs0 = gs1 = gs2 = gqr = gqx = gqy = gqz = 0.f;
            drgb[0] = drgb[1] = drgb[2] = 0.f;
#pragma unroll
            for (int t = 0; t < 6; ++t) gc[t] = 0.f;
        }
    }
    float cp0 = 0.f, cp1 = 0.f, cp2 = 0.f;            // dL/dcampos of this Gaussian

    if (live) {
        // ---- colour
        if (!colors_precomp) {
            const u32 cl = __float_as_uint(mj[9]);
            if (cl & 1u) drgb[0] = 0.f;
            if (cl & 2u) drgb[1] = 0.f;
            if (cl & 4u) drgb[2] = 0.f;
            // factored SH gradient (BagsBackwardArgs.grad_dldc): the row basis x dL/dcolour is formed later, for all views of the step at once
            if (g_dldc) { g_dldc[3 * (size_t)i] = drgb[0]; g_dldc[3 * (size_t)i + 1] = drgb[1]; g_dldc[3 * (size_t)i + 2] = drgb[2]; }
            const float ex = x - cpx, ey = y - cpy, ez = z - cpz;
            const float il = 1.0f / sqrtf(ex * ex + ey * ey + ez * ez);
            const float ux_ = ex * il, uy_ = ey * il, uz_ = ez * il;
            float bs[16];
            sh_basis(deg, ux_, uy_, uz_, bs);
            const int nb = (deg + 1) * (deg + 1);
            // gradient rows: one (M,3) row of g_shs, or (BagsInputs.shs_rest) the DC triple in g_shs + an (M-1,3) row of g_shs_rest
            float* gsh = g_shs ? (g_shs_rest ? g_shs + 3 * (size_t)i : g_shs + (size_t)i * M * 3) : nullptr;
            float* gsr = g_shs ? (g_shs_rest ? g_shs_rest + (size_t)i * (M - 1) * 3 : gsh + 3) : nullptr;
            // dL/d(direction) = M^T dL/dcolour, M from the forward (clamped channels carry a zero dL/dcolour)
            const float ddx = mj[0] * drgb[0] + mj[1] * drgb[1] + mj[2] * drgb[2];
            const float ddy = mj[3] * drgb[0] + mj[4] * drgb[1] + mj[5] * drgb[2];
            const float ddz = mj[6] * drgb[0] + mj[7] * drgb[1] + mj[8] * drgb[2];
            // dL/dsh[t][c] = basis_t dL/dcolour_c: no input row needed
            if (M == 16) {
                if (stage) {
                    float4* d4 = reinterpret_cast<float4*>(&srow[threadIdx.x][0]);
                    d4[0] = make_float4(bs[0], nb > 1 ? bs[1] : 0.f, nb > 1 ? bs[2] : 0.f, nb > 1 ? bs[3] : 0.f);
                    d4[1] = make_float4(nb > 4 ? bs[4] : 0.f, nb > 4 ? bs[5] : 0.f, nb > 4 ? bs[6] : 0.f, nb > 4 ? bs[7] : 0.f);
                    d4[2] = make_float4(nb > 4 ? bs[8] : 0.f, nb > 9 ? bs[9] : 0.f, nb > 9 ? bs[10] : 0.f, nb > 9 ? bs[11] : 0.f);
                    d4[3] = make_float4(nb > 9 ? bs[12] : 0.f, nb > 9 ? bs[13] : 0.f, nb > 9 ? bs[14] : 0.f, nb > 9 ? bs[15] : 0.f);
                    d4[4] = make_float4(drgb[0], drgb[1], drgb[2], 0.f);
                }
            } else if (gsh)
            for (int t = 0; t < M; ++t) {
                const bool on = t < nb;
                const float o0 = on ? bs[t] * drgb[0] : 0.f, o1 = on ? bs[t] * drgb[1] : 0.f, o2 = on ? bs[t] * drgb[2] : 0.f;
                float* gr = (t == 0) ? gsh : gsr + 3 * (t - 1);
                if (ACCUM) { gr[0] += o0; gr[1] += o1; gr[2] += o2; }
                else { gr[0] = o0; gr[1] = o1; gr[2] = o2; }
            }
            const float dot = ux_ * ddx + uy_ * ddy + uz_ * ddz;
            const float px_ = (ddx - ux_ * dot) * il, py_ = (ddy - uy_ * dot) * il, pz_ = (ddz - uz_ * dot) * il;
            dmx += px_; dmy += py_; dmz += pz_;
            cp0 = -px_; cp1 = -py_; cp2 = -pz_;
        }
    } else if (i < P && g_dldc && !colors_precomp) {      // a culled Gaussian: no colour gradient
        g_dldc[3 * (size_t)i] = 0.f; g_dldc[3 * (size_t)i + 1] = 0.f; g_dldc[3 * (size_t)i + 2] = 0.f;
    } else if (i < P && g_shs && !colors_precomp) {       // a culled Gaussian: its gradient row is zero
        float* gsh = g_shs_rest ? g_shs + 3 * (size_t)i : g_shs + (size_t)i * M * 3;
        float* gsr = g_shs_rest ? g_shs_rest + (size_t)i * (M - 1) * 3 : gsh + 3;
        if (M == 16) {
            if (stage) {
                float4* d4 = reinterpret_cast<float4*>(&srow[threadIdx.x][0]);
#pragma unroll
                for (int t = 0; t < 5; ++t) d4[t] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        } else if (!ACCUM) {
            gsh[0] = gsh[1] = gsh[2] = 0.f;
            for (int t = 0; t < 3 * (M - 1); ++t) gsr[t] = 0.f;
        }
    }
    if (stage && g_shs_rest) {                       // ... as two spans: 256 DC triples (3 KB) and 256 rows of 45 floats (45 KB)
        __syncthreads();
        auto span = [&](float* __restrict__ out, const u32 R, const u32 t0, const u32 rounds) {      // R floats per row, first coefficient t0
            const size_t first = (size_t)blockIdx.x * 256u * R, total = (size_t)P * R;
            for (u32 k = 0; k < rounds; ++k) {
                const u32 el = k * 256u + threadIdx.x;           // float4 number inside the workgroup's span
                if (el * 4u >= 256u * R) break;
                float o4[4]; size_t gi[4];
#pragma unroll
                for (u32 u = 0; u < 4; ++u) {
                    const u32 f = el * 4u + u, row = f / R, r = f - row * R, t = t0 + r / 3u, c = r - 3u * (r / 3u);
                    o4[u] = srow[row][t] * srow[row][16u + c];
                    gi[u] = first + f;
                }
                if (gi[3] < total) {
                    float4* d = reinterpret_cast<float4*>(out + gi[0]);
                    float4 o = make_float4(o4[0], o4[1], o4[2], o4[3]);
                    if (ACCUM) { const float4 old = *d; o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w; }
                    *d = o;
                } else {
#pragma unroll
                    for (u32 u = 0; u < 4; ++u)
                        if (gi[u] < total) out[gi[u]] = ACCUM ? out[gi[u]] + o4[u] : o4[u];
                }
            }
        };
        span(g_shs, 3u, 0u, 1u);
        span(g_shs_rest, 45u, 1u, 12u);
    } else if (stage) {                              // the workgroup's gradient rows leave as whole lines
        __syncthreads();
        float4* g4g = reinterpret_cast<float4*>(g_shs);
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            const u32 el = (u32)k * 256u + threadIdx.x;          // float4 number inside the workgroup's span
            const size_t e = base4 + el;
            if (e < lim4) {
                const u32 row = el / 12u, j = el - row * 12u;     // Gaussian of the workgroup, float4 of its row
                const float* sr = &srow[row][0];
                // floats 4j .. 4j+3 of the row: float f is coefficient f / 3, channel f % 3 (plain indexed LDS reads: a three-way
                // select on 4j % 3 came out of the compiler with one branch using the wrong operand)
                float o4[4];
#pragma unroll
                for (u32 u = 0; u < 4; ++u) {
                    const u32 fl = 4u * j + u, t = fl / 3u, c = fl - 3u * t;
                    o4[u] = sr[t] * sr[16u + c];
                }
                float4 o = make_float4(o4[0], o4[1], o4[2], o4[3]);
                if (ACCUM) { const float4 old = g4g[e]; o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w; }
                g4g[e] = o;
            }
        }
    }

    if (i < P) {
        if (ACCUM) {                                     // += into the caller's running sums
            if (g_means3D) { dmx += g_means3D[3 * i]; dmy += g_means3D[3 * i + 1]; dmz += g_means3D[3 * i + 2]; }
            if (g_opac) gop += g_opac[i];
            if (g_colors) { drgb[0] += g_colors[3 * i]; drgb[1] += g_colors[3 * i + 1]; drgb[2] += g_colors[3 * i + 2]; }
            if (g_scales) { gs0 += g_scales[3 * i]; gs1 += g_scales[3 * i + 1]; gs2 += g_scales[3 * i + 2]; }
            if (g_rot) { const float4 old = reinterpret_cast<const float4*>(g_rot)[i]; gqr += old.x; gqx += old.y; gqy += old.z; gqz += old.w; }
            if (g_cov3D) {
#pragma unroll
                for (int t = 0; t < 6; ++t) gc[t] += g_cov3D[6 * (size_t)i + t];
            }
        }
        if (g_means3D) { g_means3D[3 * i] = dmx; g_means3D[3 * i + 1] = dmy; g_means3D[3 * i + 2] = dmz; }
        if (g_means2D) { g_means2D[3 * i] = gm2x; g_means2D[3 * i + 1] = gm2y; g_means2D[3 * i + 2] = 0.f; }
        if (g_densify) { g_densify[3 * i] = gdx; g_densify[3 * i + 1] = gdy; g_densify[3 * i + 2] = 0.f; }
        if (g_opac) g_opac[i] = gop;
        if (g_colors) { g_colors[3 * i] = drgb[0]; g_colors[3 * i + 1] = drgb[1]; g_colors[3 * i + 2] = drgb[2]; }
        if (g_scales) { g_scales[3 * i] = gs0; g_scales[3 * i + 1] = gs1; g_scales[3 * i + 2] = gs2; }
        if (g_rot) reinterpret_cast<float4*>(g_rot)[i] = make_float4(gqr, gqx, gqy, gqz);
        if (g_cov3D) {
#pragma unroll
            for (int t = 0; t < 6; ++t) g_cov3D[6 * (size_t)i + t] = gc[t];
        }
    }

    // ---- 5b. the view-direction part (campos), then workgroup -> slab row
    {
        const float r0 = wave_total_f(cp0), r1 = wave_total_f(cp1), r2 = wave_total_f(cp2);
        if (lane == 63) { wpose[wave][29] = r0; wpose[wave][30] = r1; wpose[wave][31] = r2; }
    }
    __syncthreads();
    if (threadIdx.x < POSE_VALS) {
        const int t = threadIdx.x;
        float r = (t < 35) ? (wpose[0][t] + wpose[1][t]) + (wpose[2][t] + wpose[3][t]) : 0.f;
        if (t < 9) r += (wpose[0][35 + t] + wpose[1][35 + t]) + (wpose[2][35 + t] + wpose[3][35 + t]);
        pose_slab[(size_t)blockIdx.x * POSE_VALS + t] = r;
    }
}

// rows -> the five pose tensors, summed in fp64.  One workgroup per slab column (35 used): thread t adds rows t, t + 256, ...
// in row order, the 64 partial sums of a wave are added by a fixed shuffle tree, the four wave sums in wave order, so the
// result is deterministic.  (One 1024-thread workgroup for all 40 columns took 10.5 us: 78 dependent-latency-bound loads
// per thread; here every thread has its 8 loads in flight at once.)
__global__ void __launch_bounds__(256)
pose_reduce_kernel(const float* __restrict__ slab, int nblocks, float* __restrict__ g_view, float* __restrict__ g_proj,
                   float* __restrict__ g_intr, float* __restrict__ g_campos, float* __restrict__ g_shift)
{
    __shared__ double wsum[4];
    const int t = blockIdx.x;                            // slab column
    double acc = 0.0;
    int b = threadIdx.x;
    // (rows past the end INSIDE the batch, read at a clamped index and dropped afterwards: a remainder loop of single loads -- 1954 rows are
    // 7.6 per thread, so most threads took it -- was seven dependent round trips, 6.5 us of a kernel that needs one.  No `in range ? load : 0`:
    // the compiler makes that a branch per load with a wait behind each.  The order of the additions is unchanged.)
    for (; b < nblocks; b += 8 * 256) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = slab[(size_t)min(b + u * 256, nblocks - 1) * POSE_VALS + t];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += (b + u * 256 < nblocks) ? (double)v[u] : 0.0;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x != 0) return;
    const float val = (float)(((wsum[0] + wsum[1]) + wsum[2]) + wsum[3]);
    pose_write_out(t, val, g_view, g_proj, g_intr, g_campos, g_shift);
}

hipError_t launch_preprocess_bwd(const BagsSettings& s, const BagsInputs& in, const GeomView& g, const int32_t*,
                                 const float* partials_records, float* pose_slab, int* nblocks_out, const BagsBackwardArgs& a,
                                 hipStream_t st, bool binned, const unsigned char* live_map)
{
    const int P = in.P;
    const int nb = cdiv(P, 256);
    *nblocks_out = nb;
    if (P == 0) return hipSuccess;
    const float* partials = partials_records;
#define PRE_BWD_LAUNCH(COV)     LAUNCH_K((preprocess_bwd_kernel<COV, ACC_, LIVE_>), dim3(nb), dim3(256), 0, st, P, s.sh_coeffs, s.sh_degree, s.image_width, \
                       s.image_height, s.tanfovx, s.tanfovy, s.scale_modifier, (s.clamp_grad == BAGS_CLAMP_GRAD_EXACT) ? 0 : 1, \
                       (s.conic_grad == BAGS_CONIC_GRAD_EXACT) ? 0 : 1, in.means3D, in.shift_factors, in.shs, \
                       in.colors_precomp, in.scales, in.rotations, in.cov3D_precomp, s.viewmatrix, s.projmatrix, \
                       s.intrinsic, s.campos, in.opacities, g.rec_count, binned ? nullptr : g.inst_off, g.local_off, g.block_base, \
                       binned_per_block(P), g.shjac, partials, live_map, \
                       pose_slab, a.grad_means3D, a.grad_means2D, a.grad_means2D_densify, a.grad_shs, in.shs_rest ? a.grad_shs_rest : nullptr, \
                       (in.shs && !in.colors_precomp) ? a.grad_dldc : nullptr, \
                       a.grad_colors_precomp, a.grad_opacities, a.grad_scales, a.grad_rotations, a.grad_cov3D_precomp);
#define PRE_BWD_PICK if (in.cov3D_precomp) { PRE_BWD_LAUNCH(true) } else { PRE_BWD_LAUNCH(false) }
    if (live_map) {
#define LIVE_ true
        if (a.accumulate) {
#define ACC_ true
            PRE_BWD_PICK
#undef ACC_
        } else {
#define ACC_ false
            PRE_BWD_PICK
#undef ACC_
        }
#undef LIVE_
    } else {
#define LIVE_ false
        if (a.accumulate) {
#define ACC_ true
            PRE_BWD_PICK
#undef ACC_
        } else {
#define ACC_ false
            PRE_BWD_PICK
#undef ACC_
        }
#undef LIVE_
    }
#undef PRE_BWD_PICK
#undef PRE_BWD_LAUNCH
    return hipGetLastError();
}

// ---- (ABI 10) SH-gradient rows of several views from their factored form.  One 256-thread workgroup per SHV_G = 128 Gaussians, view
// after view: threads 0..127 form their Gaussian's basis at the view's direction (the very expressions preprocess_bwd_kernel uses: same
// file, same flags, same bits) and stage 16 + 3 floats; then the workgroup's 128 rows leave as whole lines, thread t owning float4
// number k * 256 + t of the span for ALL views -- the running sums of a step live in registers (six or seven float4s per thread: with
// 256 Gaussians per workgroup the thirteen float4s and their index arithmetic did not fit 128 registers), the rows are written once.
#define SHV_G 128
struct ShViewPtrs { const float* campos[BAGS_MAX_SH_VIEWS]; const float* dldc[BAGS_MAX_SH_VIEWS]; };
template <bool SPLIT>
__global__ void __launch_bounds__(256, SPLIT ? 3 : 4)      // (split: 45-float rows, dearer index arithmetic: 19 spilled registers at four per CU)
sh_grad_from_views_kernel(int P, int deg, const float* __restrict__ means3D, const ShViewPtrs V, int n_views,
                          float* __restrict__ g_shs, float* __restrict__ g_shs_rest, int accumulate)
{
    __shared__ float srow[SHV_G][20];
    const int i = blockIdx.x * SHV_G + (threadIdx.x & (SHV_G - 1));
    const size_t ic = (size_t)(i < P ? i : P - 1);
    const float x = means3D[3 * ic], y = means3D[3 * ic + 1], z = means3D[3 * ic + 2];
    const int nb = (deg + 1) * (deg + 1);
    // the float4s of the workgroup's span(s) this thread owns: concatenated (P,16,3): 6 of one span of 128 x 48 floats; split: 1 of the DC
    // span (128 x 3 floats = 96 float4s: threads 0..95) + 6 of the (P,15,3) span (128 x 45 floats = 1440 float4s: the last round partial)
    constexpr int NK = SPLIT ? 7 : 6;
    float4 acc[NK];
    bool first = (accumulate == 0);
    for (int v = 0; v < n_views; ++v) {
        __syncthreads();                                  // the previous view's rows have been consumed
        if (threadIdx.x < SHV_G) {
            const float cpx = V.campos[v][0], cpy = V.campos[v][1], cpz = V.campos[v][2];
            const float* dl = V.dldc[v] + 3 * ic;
            const float d0 = dl[0], d1 = dl[1], d2 = dl[2];
            const float ex = x - cpx, ey = y - cpy, ez = z - cpz;
            const float il = 1.0f / sqrtf(ex * ex + ey * ey + ez * ez);
            const float ux_ = ex * il, uy_ = ey * il, uz_ = ez * il;
            float bs[16];
            sh_basis(deg, ux_, uy_, uz_, bs);
            float4* d4 = reinterpret_cast<float4*>(&srow[threadIdx.x][0]);
            const bool on = i < P;
            d4[0] = make_float4(bs[0], nb > 1 ? bs[1] : 0.f, nb > 1 ? bs[2] : 0.f, nb > 1 ? bs[3] : 0.f);
            d4[1] = make_float4(nb > 4 ? bs[4] : 0.f, nb > 4 ? bs[5] : 0.f, nb > 4 ? bs[6] : 0.f, nb > 4 ? bs[7] : 0.f);
            d4[2] = make_float4(nb > 4 ? bs[8] : 0.f, nb > 9 ? bs[9] : 0.f, nb > 9 ? bs[10] : 0.f, nb > 9 ? bs[11] : 0.f);
            d4[3] = make_float4(nb > 9 ? bs[12] : 0.f, nb > 9 ? bs[13] : 0.f, nb > 9 ? bs[14] : 0.f, nb > 9 ? bs[15] : 0.f);
            d4[4] = make_float4(on ? d0 : 0.f, on ? d1 : 0.f, on ? d2 : 0.f, 0.f);
        }
        __syncthreads();
        // products of this view for the float4s the thread owns; the first view starts the sums (or adds to the buffers' content)
        auto add = [&](const int k, const u32 R, const u32 t0, const u32 el, float* __restrict__ out, const size_t first_f, const size_t total) {
            float o4[4];
#pragma unroll
            for (u32 u = 0; u < 4; ++u) {
                const u32 f = el * 4u + u, row = f / R, r = f - row * R, t = t0 + r / 3u, c = r - 3u * (r / 3u);
                o4[u] = srow[row][t] * srow[row][16u + c];
            }
            if (first) acc[k] = make_float4(o4[0], o4[1], o4[2], o4[3]);
            else {
                if (v == 0) {                                // accumulate: the running sums start from what the buffers hold
                    const size_t g0 = first_f + (size_t)el * 4u;
                    if (g0 + 3 < total) acc[k] = *reinterpret_cast<const float4*>(out + g0);
                    else acc[k] = make_float4(g0 < total ? out[g0] : 0.f, g0 + 1 < total ? out[g0 + 1] : 0.f, g0 + 2 < total ? out[g0 + 2] : 0.f, 0.f);
                }
                acc[k].x = o4[0] + acc[k].x; acc[k].y = o4[1] + acc[k].y; acc[k].z = o4[2] + acc[k].z; acc[k].w = o4[3] + acc[k].w;
            }
        };
        if (SPLIT) {
            if (threadIdx.x < (u32)(SHV_G * 3 / 4)) add(0, 3u, 0u, threadIdx.x, g_shs, (size_t)blockIdx.x * SHV_G * 3u, (size_t)P * 3u);
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const u32 el = (u32)k * 256u + threadIdx.x;
                if (el * 4u < (u32)SHV_G * 45u) add(1 + k, 45u, 1u, el, g_shs_rest, (size_t)blockIdx.x * SHV_G * 45u, (size_t)P * 45u);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 6; ++k) add(k, 48u, 0u, (u32)k * 256u + threadIdx.x, g_shs, (size_t)blockIdx.x * SHV_G * 48u, (size_t)P * 48u);
        }
        first = false;
    }
    auto put = [&](const int k, const u32 el, float* __restrict__ out, const size_t first_f, const size_t total) {
        const size_t g0 = first_f + (size_t)el * 4u;
        if (g0 + 3 < total) *reinterpret_cast<float4*>(out + g0) = acc[k];
        else {
            if (g0 < total) out[g0] = acc[k].x;
            if (g0 + 1 < total) out[g0 + 1] = acc[k].y;
            if (g0 + 2 < total) out[g0 + 2] = acc[k].z;
        }
    };
    if (n_views <= 0) return;
    if (SPLIT) {
        if (threadIdx.x < (u32)(SHV_G * 3 / 4)) put(0, threadIdx.x, g_shs, (size_t)blockIdx.x * SHV_G * 3u, (size_t)P * 3u);
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const u32 el = (u32)k * 256u + threadIdx.x;
            if (el * 4u < (u32)SHV_G * 45u) put(1 + k, el, g_shs_rest, (size_t)blockIdx.x * SHV_G * 45u, (size_t)P * 45u);
        }
    } else {
#pragma unroll
        for (int k = 0; k < 6; ++k) put(k, (u32)k * 256u + threadIdx.x, g_shs, (size_t)blockIdx.x * SHV_G * 48u, (size_t)P * 48u);
    }
}
// any other M (fewer stored coefficients): a thread per Gaussian writes its own row; the same products in the same order
__global__ void __launch_bounds__(256)
sh_grad_from_views_generic_kernel(int P, int M, int deg, const float* __restrict__ means3D, const ShViewPtrs V, int n_views,
                                  float* __restrict__ g_shs, float* __restrict__ g_shs_rest, int accumulate)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P || n_views <= 0) return;
    const float x = means3D[3 * (size_t)i], y = means3D[3 * (size_t)i + 1], z = means3D[3 * (size_t)i + 2];
    const int nb = (deg + 1) * (deg + 1);
    float* gsh = g_shs_rest ? g_shs + 3 * (size_t)i : g_shs + (size_t)i * M * 3;
    float* gsr = g_shs_rest ? g_shs_rest + (size_t)i * (M - 1) * 3 : gsh + 3;
    for (int v = 0; v < n_views; ++v) {
        const float cpx = V.campos[v][0], cpy = V.campos[v][1], cpz = V.campos[v][2];
        const float* dl = V.dldc[v] + 3 * (size_t)i;
        const float d[3] = {dl[0], dl[1], dl[2]};
        const float ex = x - cpx, ey = y - cpy, ez = z - cpz;
        const float il = 1.0f / sqrtf(ex * ex + ey * ey + ez * ez);
        float bs[16];
        sh_basis(deg, ex * il, ey * il, ez * il, bs);
        const bool add = accumulate != 0 || v > 0;
        for (int t = 0; t < M; ++t) {
            float* gr = (t == 0) ? gsh : gsr + 3 * (t - 1);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float o = (t < nb) ? bs[t] * d[c] : 0.f;
                gr[c] = add ? o + gr[c] : o;
            }
        }
    }
}
hipError_t launch_sh_grad_from_views(int P, int M, int deg, const float* means3D, const BagsShViews& views, float* g_shs, float* g_shs_rest,
                                     int accumulate, hipStream_t st)
{
    if (P == 0 || views.n_views <= 0) return hipSuccess;
    ShViewPtrs V;
    for (int v = 0; v < BAGS_MAX_SH_VIEWS; ++v) { V.campos[v] = views.campos[v < views.n_views ? v : 0]; V.dldc[v] = views.dldc[v < views.n_views ? v : 0]; }
    const int nbk = cdiv(P, 256), nbs = cdiv(P, SHV_G);
    if (M == 16 && g_shs_rest) hipLaunchKernelGGL(sh_grad_from_views_kernel<true>, dim3(nbs), dim3(256), 0, st, P, deg, means3D, V, views.n_views, g_shs, g_shs_rest, accumulate);
    else if (M == 16) hipLaunchKernelGGL(sh_grad_from_views_kernel<false>, dim3(nbs), dim3(256), 0, st, P, deg, means3D, V, views.n_views, g_shs, g_shs_rest, accumulate);
    else hipLaunchKernelGGL(sh_grad_from_views_generic_kernel, dim3(nbk), dim3(256), 0, st, P, M, deg, means3D, V, views.n_views, g_shs, g_shs_rest, accumulate);
    return hipGetLastError();
}

hipError_t launch_pose_reduce(const float* pose_slab, int nblocks, const BagsBackwardArgs& a, hipStream_t st)
{
    LAUNCH_K(pose_reduce_kernel, dim3(35), dim3(256), 0, st, pose_slab, nblocks, a.grad_viewmatrix,
                       a.grad_projmatrix, a.grad_intrinsic, a.grad_campos, a.grad_shift_factors);
    return hipGetLastError();
}
