// preprocess_fwd.hip -- K1: per-Gaussian 3D->2D projection, EWA covariance, radius, tile rectangle, SH colour.
//
// COMPILED WITH -ffp-contract=off: the integer artefacts (radius, tile rectangle, depth key) must be bit-identical
// to the fp32 oracle (oracle/raster_oracle.py: preprocess), so every expression here is the same sequence of
// individually rounded IEEE fp32 mul/add/div/sqrt, left to right.  Algorithm: SURVEY.md Appendix A.1; conventions:
// utils/graphics_utils.py:26-33 (row-vector transforms), utils/general_utils.py:130-163 (quaternion order, R.S),
// utils/sh_utils.py:57-112 (SH basis), gaussian_renderer/__init__.py:92-95 (+0.5, clamp at 0).
#include "bags_common.h"
#include "sh_basis.h"
#include "binning_common.h"

struct CamConst {
    float v[16], m[16], k[16];
    float campos[3];
    float sf[3];
};

// D7, second half: which tiles of a rectangle of w x h <= 8 x 8 tiles the ellipse Q(d) = a dx^2 + 2 b dx dy + c dy^2 <= tau2m
// around (px, py) reaches.  The minimum of the convex Q over the square of a tile's pixel centres [X0, X0 + 15] x [Y0, Y0 +
// 15] is 0 for a centre inside it; otherwise it lies on an edge that faces the centre, at the 1-D minimiser along that edge
// clamped to the edge.  tau2m carries the margins (1e-3 + 0.02 on 2 ln, then 2 %): the blend kernels evaluate the same Q
// from the same a, b, c, px, py with offsets of at most 8 tiles, so their rounding (~1e-6 Q) cannot cross it -- a dropped
// tile holds no pixel with alpha >= 1/255, the image and every gradient are those of the full rectangle.
// Adds, multiplies, two reciprocals, min/max only, contraction off: oracle/raster_oracle.py:_tile_reach repeats them in
// torch and gets the same bits.  Bit ry * 8 + rx of the result = tile (ex0 + rx, ey0 + ry) is emitted.
__device__ __forceinline__ u64 tile_reach(float px, float py, float a, float b, float c, float tau2m, int ex0, int ey0, int w, int h)
{
    const float ra = 1.0f / a, rc = 1.0f / c, b2 = 2.0f * b;
    u64 keep = 0ull;
    for (int ry = 0; ry < h; ++ry) {
        const float Y0 = (float)((ey0 + ry) * BAGS_TILE), Y1 = Y0 + 15.0f;
        const bool yin = (py >= Y0) && (py <= Y1);
        const float dyE = ((py < Y0) ? Y0 : Y1) - py;                  // the horizontal edge that faces the centre (if !yin)
        const float xs = px - (b * dyE) * ra;                           // unconstrained minimiser along it
        const float cdd = (c * dyE) * dyE;
        for (int rx = 0; rx < w; ++rx) {
            const float X0 = (float)((ex0 + rx) * BAGS_TILE), X1 = X0 + 15.0f;
            const bool xin = (px >= X0) && (px <= X1);
            const float dx = fminf(fmaxf(xs, X0), X1) - px;
            const float qh = (a * dx) * dx + (b2 * dx) * dyE + cdd;
            const float dxE = ((px < X0) ? X0 : X1) - px;               // the vertical edge that faces the centre (if !xin)
            const float ys = py - (b * dxE) * rc;
            const float dy = fminf(fmaxf(ys, Y0), Y1) - py;
            const float qv = (a * dxE) * dxE + (b2 * dxE) * dy + (c * dy) * dy;
            const float q = fminf(yin ? INFINITY : qh, xin ? INFINITY : qv);
            if ((xin && yin) || !(q > tau2m)) keep |= 1ull << (ry * 8 + rx);
        }
    }
    return keep;
}

// Everything K1 computes for one Gaussian, as values: the two kernels below differ in what happens between the arithmetic and
// the stores (nothing / the (block, tile) count and the block-local instance offset).
struct K1Args {
    int P, M, deg, W, H;
    float tanfovx, tanfovy, mod;
    int depth_mode, tile_bounds;
    int rec_opacity;         // tile-binned path with the stock tile rule: the LISTS follow the stock 3-sigma square, the partial-gradient
                             // RECORDS only exist for the tiles the alpha >= 1/255 ellipse reaches (the opacity rule's rectangle + tile mask)
    const float *means3D, *means2D, *shs, *colors_precomp, *opacities, *scales, *rotations, *cov3D_precomp;
    const float* shs_rest;   // BagsInputs.shs_rest: `shs` is then the (P,1,3) DC tensor and this the (P,M-1,3) rest (no concatenation upstream)
    float* shjac;
};
struct K1Result {
    u32 key, tiles; int radius; uint2 rect; u64 keep; float2 pxy; float4 q0, rgbz_v;
    u32 rtiles; uint2 rrect; u64 rkeep;          // the tiles a partial-gradient record exists for (= tiles / rect / keep unless rec_opacity)
};
struct K1Outputs {
    u32* depth_key; float4* g2d; uint2* rect; u32* tiles_touched; u64* keep; int32_t* radii; float* mean2D; u32* rec_count;
};

template <bool SPLIT>            // SPLIT: BagsInputs.shs_rest given (a template parameter: as a run-time branch around the two ways of
__device__ __forceinline__ K1Result k1_project(const K1Args& A, const CamConst& cam, const int i)     // filling c[48] it cost 22-50 spilled registers)
{
    const int M = A.M, deg = A.deg, W = A.W, H = A.H, depth_mode = A.depth_mode, tile_bounds = A.tile_bounds;
    const bool want_opacity = (tile_bounds == BAGS_TILES_OPACITY) || (A.rec_opacity != 0);
    u32 rtiles = 0; uint2 rrect = make_uint2(0u, 0u); u64 rkeep = ~0ull;
    const float tanfovx = A.tanfovx, tanfovy = A.tanfovy, mod = A.mod;
    const float* __restrict__ means3D = A.means3D; const float* __restrict__ means2D = A.means2D;
    const float* __restrict__ shs = A.shs; const float* __restrict__ colors_precomp = A.colors_precomp;
    const float* __restrict__ opacities = A.opacities; const float* __restrict__ scales = A.scales;
    const float* __restrict__ rotations = A.rotations; const float* __restrict__ cov3D_precomp = A.cov3D_precomp;
    float* __restrict__ shjac = A.shjac;
    const float* v = cam.v; const float* m = cam.m; const float* k = cam.k;

    // defaults for a culled Gaussian
    u32 key = KEY_CULLED; u32 tiles = 0; int radius = 0;
    uint2 rect = make_uint2(0u, 0u);
    u64 keep = ~0ull; bool masked = false;
    float2 pxy = make_float2(0.f, 0.f);
    float4 q0 = make_float4(0.f, 0.f, 0.f, 0.f), rgbz_v = q0;

    const float x = means3D[3 * i + 0], y = means3D[3 * i + 1], z = means3D[3 * i + 2];
    const float tx = x * v[0] + y * v[4] + z * v[8] + v[12];
    const float ty = x * v[1] + y * v[5] + z * v[9] + v[13];
    const float tz = x * v[2] + y * v[6] + z * v[10] + v[14];
    bool ok = tz > 0.2f;                                   // near-plane cull
    if (ok) {
        // D2: entrance-pupil shift (zero factors => exact identity)
        const float rho = sqrtf(tx * tx + ty * ty + 1e-20f);
        const float theta = det_atan2_pos(rho, tz);
        const float th2 = theta * theta;
        const float th3 = th2 * theta;
        const float shift = cam.sf[0] * th3 + cam.sf[1] * (th3 * th2) + cam.sf[2] * (th3 * th2 * th2);
        const float tzs = tz + shift;

        const float hx = x * m[0] + y * m[4] + z * m[8] + m[12] + shift * k[8];
        const float hy = x * m[1] + y * m[5] + z * m[9] + m[13] + shift * k[9];
        const float hw = x * m[3] + y * m[7] + z * m[11] + m[15] + shift * k[11];
        const float pw = 1.0f / (hw + 1e-7f);
        float ndc_x = hx * pw, ndc_y = hy * pw;
        if (means2D) { ndc_x = ndc_x + means2D[3 * i + 0]; ndc_y = ndc_y + means2D[3 * i + 1]; }
        const float px = ((ndc_x + 1.0f) * (float)W - 1.0f) * 0.5f;
        const float py = ((ndc_y + 1.0f) * (float)H - 1.0f) * 0.5f;

        float c0, c1, c2, c3, c4, c5;
        if (cov3D_precomp) {
            const float* c = cov3D_precomp + 6 * (size_t)i;
            c0 = c[0]; c1 = c[1]; c2 = c[2]; c3 = c[3]; c4 = c[4]; c5 = c[5];
        } else {
            const float s0 = scales[3 * i + 0] * mod, s1 = scales[3 * i + 1] * mod, s2 = scales[3 * i + 2] * mod;
            const float4 q = reinterpret_cast<const float4*>(rotations)[i];
            const float qr = q.x, qx = q.y, qy = q.z, qz = q.w;
            const float r00 = 1.0f - 2.0f * (qy * qy + qz * qz);
            const float r01 = 2.0f * (qx * qy - qr * qz);
            const float r02 = 2.0f * (qx * qz + qr * qy);
            const float r10 = 2.0f * (qx * qy + qr * qz);
            const float r11 = 1.0f - 2.0f * (qx * qx + qz * qz);
            const float r12 = 2.0f * (qy * qz - qr * qx);
            const float r20 = 2.0f * (qx * qz - qr * qy);
            const float r21 = 2.0f * (qy * qz + qr * qx);
            const float r22 = 1.0f - 2.0f * (qx * qx + qy * qy);
            const float l00 = r00 * s0, l01 = r01 * s1, l02 = r02 * s2;
            const float l10 = r10 * s0, l11 = r11 * s1, l12 = r12 * s2;
            const float l20 = r20 * s0, l21 = r21 * s1, l22 = r22 * s2;
            c0 = l00 * l00 + l01 * l01 + l02 * l02;
            c1 = l00 * l10 + l01 * l11 + l02 * l12;
            c2 = l00 * l20 + l01 * l21 + l02 * l22;
            c3 = l10 * l10 + l11 * l11 + l12 * l12;
            c4 = l10 * l20 + l11 * l21 + l12 * l22;
            c5 = l20 * l20 + l21 * l21 + l22 * l22;
        }
        const float fx = k[0] * (0.5f * (float)W);          // D1
        const float fy = k[5] * (0.5f * (float)H);
        const float limx = 1.3f * tanfovx, limy = 1.3f * tanfovy;
        const float txtz = tx / tzs, tytz = ty / tzs;
        const float cx_ = fminf(limx, fmaxf(-limx, txtz)) * tzs;
        const float cy_ = fminf(limy, fmaxf(-limy, tytz)) * tzs;
        const float itz = 1.0f / tzs;
        const float itz2 = itz * itz;
        const float j00 = fx * itz;
        const float j02 = -(fx * cx_) * itz2;
        const float j11 = fy * itz;
        const float j12 = -(fy * cy_) * itz2;
        const float a00 = j00 * v[0] + j02 * v[2];
        const float a01 = j00 * v[4] + j02 * v[6];
        const float a02 = j00 * v[8] + j02 * v[10];
        const float a10 = j11 * v[1] + j12 * v[2];
        const float a11 = j11 * v[5] + j12 * v[6];
        const float a12 = j11 * v[9] + j12 * v[10];
        const float b00 = a00 * c0 + a01 * c1 + a02 * c2;
        const float b01 = a00 * c1 + a01 * c3 + a02 * c4;
        const float b02 = a00 * c2 + a01 * c4 + a02 * c5;
        const float b10 = a10 * c0 + a11 * c1 + a12 * c2;
        const float b11 = a10 * c1 + a11 * c3 + a12 * c4;
        const float b12 = a10 * c2 + a11 * c4 + a12 * c5;
        const float cxx = b00 * a00 + b01 * a01 + b02 * a02 + 0.3f;
        const float cxy = b00 * a10 + b01 * a11 + b02 * a12;
        const float cyy = b10 * a10 + b11 * a11 + b12 * a12 + 0.3f;
        const float det = cxx * cyy - cxy * cxy;
        ok = (det != 0.0f);
        if (ok) {
            const float det_inv = 1.0f / det;
            const float con_a = cyy * det_inv, con_b = -cxy * det_inv, con_c = cxx * det_inv;
            const float mid = 0.5f * (cxx + cyy);
            const float lam = mid + sqrtf(fmaxf(mid * mid - det, 0.1f));
            const float rad_f = ceilf(3.0f * sqrtf(lam));
            ok = isfinite(px) && isfinite(py) && isfinite(rad_f);
            if (ok) {
                const int gx = (W + BAGS_TILE - 1) / BAGS_TILE, gy = (H + BAGS_TILE - 1) / BAGS_TILE;
                const float big = 1.0e9f;
                const int minx = min(gx, max(0, (int)fminf(big, fmaxf(-big, (px - rad_f) / 16.0f))));
                const int miny = min(gy, max(0, (int)fminf(big, fmaxf(-big, (py - rad_f) / 16.0f))));
                const int maxx = min(gx, max(0, (int)fminf(big, fmaxf(-big, (px + rad_f + 15.0f) / 16.0f))));
                const int maxy = min(gy, max(0, (int)fminf(big, fmaxf(-big, (py + rad_f + 15.0f) / 16.0f))));
                const int nt = (maxx - minx) * (maxy - miny);
                if (nt > 0) {                                // "visible" (radii > 0) is decided by the stock rectangle
                    int ex0 = minx, ey0 = miny, ex1 = maxx, ey1 = maxy;      // rectangle the instances are emitted for
                    if (want_opacity) {
                        // alpha = o exp(-d^T Q d / 2) >= 1/255 only inside d^T Q d <= 2 ln(255 o), whose axis-aligned half
                        // extents are sqrt(2 ln(255 o) cov_xx), sqrt(.. cov_yy).  ln is bounded from above with operations
                        // every IEEE implementation rounds identically (the oracle repeats them in torch): 255 o = m 2^e,
                        // ln(m) <= t - t^2/2 + t^3/3 - t^4/4 for t = m - 1 in [-1/2, 0) (all dropped terms are negative).
                        // Margins: 1e-3 + 0.02 on 2 ln, 2 % + 0.1 px on the extents (the conic the pixel test uses is the
                        // inverse of cov2D only up to det's rounding, ~1e-7 x anisotropy, a common scale on the ellipse).
                        const float visv = 255.0f * opacities[i];
                        if (!(visv >= 1.0f)) { ex1 = ex0; ey1 = ey0; }
                        else {
                            int e2; const float mant = frexpf(visv, &e2);
                            const float t = mant - 1.0f;
                            const float poly = t * (1.0f + t * (-0.5f + t * (0.33333334f + t * -0.25f)));
                            const float lnu = (float)e2 * 0.6931472f + poly + 1.0e-3f;
                            const float tau2 = 2.0f * lnu + 0.02f;
                            const float rx = sqrtf(tau2 * cxx) * 1.02f + 0.1f, ry = sqrtf(tau2 * cyy) * 1.02f + 0.1f;
                            ex0 = max(ex0, min(gx, max(0, (int)fminf(big, fmaxf(-big, (px - rx) / 16.0f)))));
                            ey0 = max(ey0, min(gy, max(0, (int)fminf(big, fmaxf(-big, (py - ry) / 16.0f)))));
                            ex1 = min(ex1, min(gx, max(0, (int)fminf(big, fmaxf(-big, (px + rx) / 16.0f)) + 1)));
                            ey1 = min(ey1, min(gy, max(0, (int)fminf(big, fmaxf(-big, (py + ry) / 16.0f)) + 1)));
                            if (ex1 <= ex0 || ey1 <= ey0) { ex1 = ex0; ey1 = ey0; }
#ifndef NO_TILE_MASKS            // experiment switch (tools/ab_masks.sh): rectangles only, as before the masks
                            else if (ex1 - ex0 <= 8 && ey1 - ey0 <= 8 && con_a > 0.0f && con_c > 0.0f &&
                                     con_a * con_c - con_b * con_b > 0.0f) {
                                keep = tile_reach(px, py, con_a, con_b, con_c, tau2 * 1.02f, ex0, ey0, ex1 - ex0, ey1 - ey0);
                                masked = true;
                            }
#endif
                        }
                    }
                    if (!masked) keep = rect_full_mask(ex1 - ex0, ey1 - ey0);
                    tiles = masked ? (u32)__popcll(keep) : (u32)((ex1 - ex0) * (ey1 - ey0));
                    radius = (int)rad_f;
                    rect = make_uint2((u32)ex0 | ((u32)ey0 << 16), (u32)ex1 | ((u32)ey1 << 16));
                    rtiles = tiles; rrect = rect; rkeep = keep;                 // records: the opacity rule's tiles
                    if (tile_bounds != BAGS_TILES_OPACITY) {                    // lists: the stock square, every tile of it
                        keep = rect_full_mask(maxx - minx, maxy - miny);
                        tiles = (u32)nt;
                        rect = make_uint2((u32)minx | ((u32)miny << 16), (u32)maxx | ((u32)maxy << 16));
                        if (!A.rec_opacity) { rtiles = tiles; rrect = rect; rkeep = keep; }
                    }
                    pxy = make_float2(px, py);
                    const float dsort = (depth_mode == BAGS_DEPTH_DISTANCE) ? sqrtf(tx * tx + ty * ty + tzs * tzs) : tzs;
                    key = __float_as_uint(dsort);
                    // colour
                    float r, g, b; u32 cl = 0;
                    if (colors_precomp) {
                        r = colors_precomp[3 * i + 0]; g = colors_precomp[3 * i + 1]; b = colors_precomp[3 * i + 2];
                    } else {
                        const float dx = x - cam.campos[0], dy = y - cam.campos[1], dz = z - cam.campos[2];
                        const float dl = sqrtf(dx * dx + dy * dy + dz * dz);
                        const float ux = dx / dl, uy = dy / dl, uz = dz / dl;
                        const int nb = (deg + 1) * (deg + 1);
                        // a Gaussian's coefficients: one (M,3) row of `shs`, or the DC triple of `shs` + an (M-1,3) row of `shs_rest`
                        const float* __restrict__ dcp = SPLIT ? shs + 3 * (size_t)i : shs + (size_t)i * M * 3;
                        const float* __restrict__ rsp = SPLIT ? A.shs_rest + (size_t)i * (M - 1) * 3 : dcp + 3;
                        r = 0.f; g = 0.f; b = 0.f;
                        // d(colour)/d(direction): the backward multiplies it with dL/dcolour and never reads the SH row again
                        float mxr = 0.f, mxg = 0.f, mxb = 0.f, myr = 0.f, myg = 0.f, myb = 0.f, mzr = 0.f, mzg = 0.f, mzb = 0.f;
                        // one coefficient at a time, its basis value and the three derivatives formed right where they are used
                        // (as arrays for all 16 coefficients they were 64 more live registers: 140 VGPRs, 3 waves per SIMD for 5)
#define SH_TERM(t, B, GX, GY, GZ)                                                                                      \
                        if ((t) < nb) {                                                                               \
                            const float k0 = CO(3 * (t)), k1 = CO(3 * (t) + 1), k2 = CO(3 * (t) + 2);                  \
                            const float bv = (B), gx_ = (GX), gy_ = (GY), gz_ = (GZ);                                  \
                            r += bv * k0; g += bv * k1; b += bv * k2;                                                  \
                            mxr += gx_ * k0; mxg += gx_ * k1; mxb += gx_ * k2;                                         \
                            myr += gy_ * k0; myg += gy_ * k1; myb += gy_ * k2;                                         \
                            mzr += gz_ * k0; mzg += gz_ * k1; mzb += gz_ * k2;                                         \
                        }
#define SH_ALL()                                                                                                       \
                        {                                                                                             \
                            const float xx = ux * ux, yy = uy * uy, zz = uz * uz, xy = ux * uy, yz = uy * uz, xz = ux * uz; \
                            SH_TERM(0, SH_C0, 0.f, 0.f, 0.f)                                                           \
                            SH_TERM(1, -SH_C1 * uy, 0.f, -SH_C1, 0.f)                                                  \
                            SH_TERM(2, SH_C1 * uz, 0.f, 0.f, SH_C1)                                                    \
                            SH_TERM(3, -SH_C1 * ux, -SH_C1, 0.f, 0.f)                                                  \
                            SH_TERM(4, SH_C2_0 * xy, SH_C2_0 * uy, SH_C2_0 * ux, 0.f)                                  \
                            SH_TERM(5, SH_C2_1 * yz, 0.f, SH_C2_1 * uz, SH_C2_1 * uy)                                  \
                            SH_TERM(6, SH_C2_2 * (2.0f * zz - xx - yy), SH_C2_2 * -2.0f * ux, SH_C2_2 * -2.0f * uy, SH_C2_2 * 4.0f * uz) \
                            SH_TERM(7, SH_C2_3 * xz, SH_C2_3 * uz, 0.f, SH_C2_3 * ux)                                  \
                            SH_TERM(8, SH_C2_4 * (xx - yy), SH_C2_4 * 2.0f * ux, SH_C2_4 * -2.0f * uy, 0.f)            \
                            SH_TERM(9, SH_C3_0 * uy * (3.0f * xx - yy), SH_C3_0 * 6.0f * xy, SH_C3_0 * (3.0f * xx - 3.0f * yy), 0.f) \
                            SH_TERM(10, SH_C3_1 * xy * uz, SH_C3_1 * yz, SH_C3_1 * xz, SH_C3_1 * xy)                   \
                            SH_TERM(11, SH_C3_2 * uy * (4.0f * zz - xx - yy), SH_C3_2 * -2.0f * xy, SH_C3_2 * (4.0f * zz - xx - 3.0f * yy), SH_C3_2 * 8.0f * yz) \
                            SH_TERM(12, SH_C3_3 * uz * (2.0f * zz - 3.0f * xx - 3.0f * yy), SH_C3_3 * -6.0f * xz, SH_C3_3 * -6.0f * yz, SH_C3_3 * (6.0f * zz - 3.0f * xx - 3.0f * yy)) \
                            SH_TERM(13, SH_C3_4 * ux * (4.0f * zz - xx - yy), SH_C3_4 * (4.0f * zz - 3.0f * xx - yy), SH_C3_4 * -2.0f * xy, SH_C3_4 * 8.0f * xz) \
                            SH_TERM(14, SH_C3_5 * uz * (xx - yy), SH_C3_5 * 2.0f * xz, SH_C3_5 * -2.0f * yz, SH_C3_5 * (xx - yy)) \
                            SH_TERM(15, SH_C3_6 * ux * (xx - 3.0f * yy), SH_C3_6 * (3.0f * xx - 3.0f * yy), SH_C3_6 * -6.0f * xy, 0.f) \
                        }
                        if (M == 16) {             // 192 B per Gaussian, 16-byte aligned: 12 dwordx4 loads, all requested at once
                            float c[48];
                            if (SPLIT) {                  // 12 + 180 bytes, the second row only 4-byte aligned (dwordx4 loads at any dword)
                                struct __attribute__((packed, aligned(4))) UF4 { float x, y, z, w; };
                                c[0] = dcp[0]; c[1] = dcp[1]; c[2] = dcp[2];
                                const UF4* u4 = reinterpret_cast<const UF4*>(rsp);
#pragma unroll
                                for (int t = 0; t < 11; ++t) {
                                    const UF4 w = u4[t];
                                    c[3 + 4 * t] = w.x; c[4 + 4 * t] = w.y; c[5 + 4 * t] = w.z; c[6 + 4 * t] = w.w;
                                }
                                c[47] = rsp[44];
                            } else {
                                const float4* s4 = reinterpret_cast<const float4*>(dcp);
#pragma unroll
                                for (int t = 0; t < 12; ++t) {
                                    float4 w = s4[t];      // (non-temporal loads here: K1 58 -> 93 us, profiles/r05/ab_k1.txt -- a lane's twelve
                                                           //  16-byte pieces share 128-byte lines, and the lines have to stay cached between them)
                                    c[4 * t] = w.x; c[4 * t + 1] = w.y; c[4 * t + 2] = w.z; c[4 * t + 3] = w.w;
                                }
                            }
#define CO(k) c[k]
                            SH_ALL()
#undef CO
                        } else {
#define CO(k) ((k) < 3 ? dcp[(k)] : rsp[(k) - 3])
                            SH_ALL()
#undef CO
                        }
#undef SH_ALL
#undef SH_TERM
                        // 9 floats = 36 bytes per Gaussian, three 12-byte stores
                        float* mj = shjac + 10 * (size_t)i;           // 9 floats + the clamp bits (written below): 40 bytes per Gaussian
                        mj[0] = mxr; mj[1] = mxg; mj[2] = mxb; mj[3] = myr; mj[4] = myg; mj[5] = myb; mj[6] = mzr; mj[7] = mzg; mj[8] = mzb;
                        r += 0.5f; g += 0.5f; b += 0.5f;
                        if (r < 0.f) { cl |= 1u; r = 0.f; }
                        if (g < 0.f) { cl |= 2u; g = 0.f; }
                        if (b < 0.f) { cl |= 4u; b = 0.f; }
                        mj[9] = __uint_as_float(cl);
                    }
                    q0 = make_float4(con_a, con_b, con_c, opacities[i]);
                    rgbz_v = make_float4(r, g, b, tzs);
                }
            }
        }
    }
    K1Result R;
    R.key = key; R.tiles = tiles; R.radius = radius; R.rect = rect; R.keep = keep; R.pxy = pxy; R.q0 = q0; R.rgbz_v = rgbz_v;
    R.rtiles = rtiles; R.rrect = rrect; R.rkeep = rkeep;
    return R;
}

// One full 64-byte line per Gaussian (the blend kernels gather it per instance) plus the compact arrays of the binning kernels.
//   q3 = (tile mask lo, block of Gaussians, instance offset inside the block, tile mask hi): blend_bwd finds a Gaussian's
//   first partial-gradient record at block_base[q3.y] + q3.z without a second gather (tile-binned path; zeros on the radix path,
//   which keeps the inst_off array).  Rectangle and tile mask in the line are those of the RECORDS (rrect / rkeep): with the
//   stock tile rule on the tile-binned path they are the opacity rule's, a subset of the tiles the lists cover -- an instance
//   whose tile is not among them has no record, and the blend kernels skip it (round 4).
__device__ __forceinline__ void k1_store(const K1Outputs& O, const int i, const K1Result& R, const u32 blk, const u32 loff)
{
    O.depth_key[i] = R.key;
    O.tiles_touched[i] = R.tiles;
    O.rect[i] = R.rect;
    O.keep[i] = R.keep;
    float4* rec = O.g2d + 4 * (size_t)i;
    rec[0] = R.q0;
    rec[1] = make_float4(R.pxy.x, R.pxy.y, R.rgbz_v.x, R.rgbz_v.y);
    rec[2] = make_float4(R.rgbz_v.z, R.rgbz_v.w, __uint_as_float(R.rrect.x), __uint_as_float(R.rrect.y));
    rec[3] = make_float4(__uint_as_float((u32)R.rkeep), __uint_as_float(blk), __uint_as_float(loff), __uint_as_float((u32)(R.rkeep >> 32)));
    O.rec_count[i] = R.rtiles;
    O.radii[i] = R.radius;
    if (O.mean2D) { O.mean2D[2 * i] = R.pxy.x; O.mean2D[2 * i + 1] = R.pxy.y; }
}

__device__ __forceinline__ void k1_load_camera(CamConst& cam, const float* __restrict__ viewmatrix, const float* __restrict__ projmatrix,
                                               const float* __restrict__ intrinsic, const float* __restrict__ campos_p,
                                               const float* __restrict__ shift_factors)
{
    // all five loads first (clamped indices, every thread: 51 distinct words), then the LDS stores: written as `cam.x[t] = ptr[t]` pairs under
    // their `if`s this was three dependent round trips at the start of a workgroup that lives as long as the whole launch
    const int t16 = min((int)threadIdx.x, 15), t3 = min((int)threadIdx.x, 2);
    const float v = viewmatrix[t16], m = projmatrix[t16], k = intrinsic[t16];
    const float c = campos_p[t3];
    const float sfv = (shift_factors ? shift_factors : campos_p)[t3];
    if (threadIdx.x < 16) { cam.v[threadIdx.x] = v; cam.m[threadIdx.x] = m; cam.k[threadIdx.x] = k; }
    if (threadIdx.x < 3) { cam.campos[threadIdx.x] = c; cam.sf[threadIdx.x] = shift_factors ? sfv : 0.0f; }
}

template <bool SPLIT>
__global__ void __launch_bounds__(256, 5)     // 5 waves per SIMD (<= 96 VGPRs): this kernel lives on occupancy
preprocess_fwd_kernel(const K1Args A, const float* __restrict__ viewmatrix, const float* __restrict__ projmatrix,
                      const float* __restrict__ intrinsic, const float* __restrict__ campos_p,
                      const float* __restrict__ shift_factors, const K1Outputs O)
{
    __shared__ CamConst cam;
    k1_load_camera(cam, viewmatrix, projmatrix, intrinsic, campos_p, shift_factors);
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A.P) return;
    k1_store(O, i, k1_project<SPLIT>(A, cam, i), 0u, 0u);
}

// K1 + step 1 of the tile-binned lists (binning.hip) in one launch (round 4).  One 1024-thread workgroup per block of
// `per_block` consecutive Gaussians (<= 256 blocks, the row index of the (block, tile) count matrix): pass p handles Gaussian
// block * per_block + p * 1024 + tid, counts its instances per tile in LDS (packed 16-bit counters, integer LDS atomics: counts
// do not depend on arrival order), and hands it a range of record slots inside the block (wave scan + one LDS counter) --
// written into its geometry line and the compact local_off array.  tile_count_kernel was a
// 9 us launch of its own on the path to the instance count (its work, 2 M LDS atomics, hides behind K1's memory traffic here),
// and the offset inside the line saves blend_bwd a 4-byte gather per instance.
template <bool SPLIT>
__global__ void __launch_bounds__(BIN_THREADS)
preprocess_fwd_count_kernel(const K1Args A, const float* __restrict__ viewmatrix, const float* __restrict__ projmatrix,
                            const float* __restrict__ intrinsic, const float* __restrict__ campos_p,
                            const float* __restrict__ shift_factors, const K1Outputs O, const int per_block, const int grid_x,
                            const int T2, u32* __restrict__ cnt_rows, u32* __restrict__ local_off, u32* __restrict__ block_total)
{
    extern __shared__ u32 cnt[];                             // T2 packed words
    __shared__ CamConst cam;
    __shared__ u32 s_run;                                    // instances handed out so far in this block
    for (int t = threadIdx.x; t < T2; t += BIN_THREADS) cnt[t] = 0u;
    k1_load_camera(cam, viewmatrix, projmatrix, intrinsic, campos_p, shift_factors);
    if (threadIdx.x == 0) s_run = 0u;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int passes = per_block / BIN_THREADS;
    // No barrier inside the loop: the 16 waves drift apart, so the loads of one overlap the arithmetic and the stores of another
    // (in lockstep -- a block-wide scan per pass -- the 245 workgroups of a 500 k scene all loaded, then all computed, then all
    // stored: +7 us).  A Gaussian's records only need a range of their own inside the block, not a particular one: each wave
    // scans its 64 counts and takes the range from an LDS counter.  Which wave gets which range varies from run to run; no
    // result depends on it (records are summed per Gaussian in tile-list order wherever they lie).
#pragma unroll 1
    for (int pass = 0; pass < passes; ++pass) {
        const long long gi = (long long)blockIdx.x * per_block + (long long)pass * BIN_THREADS + threadIdx.x;
        const bool valid = gi < (long long)A.P;
        const int i = (int)(valid ? gi : 0);
        K1Result R;
        R.key = KEY_CULLED; R.tiles = 0; R.radius = 0; R.rect = make_uint2(0u, 0u); R.keep = ~0ull; R.pxy = make_float2(0.f, 0.f);
        R.q0 = make_float4(0.f, 0.f, 0.f, 0.f); R.rgbz_v = R.q0; R.rtiles = 0; R.rrect = make_uint2(0u, 0u); R.rkeep = ~0ull;
        if (valid) R = k1_project<SPLIT>(A, cam, i);
        __builtin_amdgcn_sched_barrier(0);
        // ---- (block, tile) counts: small rectangles by their tile mask, larger ones tile by tile, huge ones by the whole wave
        const u32 nt = R.tiles;
        const uint2 rc = R.rect;
        const int w = (int)(rc.y & 0xFFFF) - (int)(rc.x & 0xFFFF), h = (int)(rc.y >> 16) - (int)(rc.x >> 16);
        if (nt > 0) {
            if (rect_small(w, h)) walk_mask<false>(cnt, rc, R.keep, grid_x, 0u, nullptr);
            else if (nt <= BIN_COOP) walk_rect<false>(cnt, rc, grid_x, lane, false, 0u, nullptr);
        }
        u64 big = __ballot(nt > BIN_COOP);                   // never a small rectangle (at most 64 tiles)
        while (big) {
            const int src = __ffsll((long long)big) - 1;
            big &= big - 1;
            const uint2 brc = make_uint2((u32)__shfl((int)rc.x, src), (u32)__shfl((int)rc.y, src));
            walk_rect<false>(cnt, brc, grid_x, lane, true, 0u, nullptr);
        }
        // ---- the wave's range of record slots inside the block (records, not list instances: see K1Args::rec_opacity)
        const u32 nrec = R.rtiles;
        const u32 incl = wave_incl_scan(nrec);
        u32 base = 0;
        if (lane == 63) base = atomicAdd(&s_run, incl);
        base = (u32)__builtin_amdgcn_readlane((int)base, 63);
        const u32 loff = base + incl - nrec;
        __builtin_amdgcn_sched_barrier(0);
        if (valid) { k1_store(O, i, R, (u32)blockIdx.x, loff); local_off[i] = loff; }
        __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();                                         // every counter of the block is final
    if (threadIdx.x == 0) block_total[blockIdx.x] = s_run;
    u32* row = cnt_rows + (size_t)blockIdx.x * T2;
    for (int t = threadIdx.x; t < T2; t += BIN_THREADS) row[t] = cnt[t];
}

hipError_t launch_preprocess_fwd(const BagsSettings& s, const BagsInputs& in, const GeomView& g, int32_t* radii,
                                 float* mean2D, hipStream_t st, const ImgView* count_into, int grid_x)
{
    const int P = in.P;
    if (P == 0) return hipSuccess;
    K1Args A;
    A.P = P; A.M = s.sh_coeffs; A.deg = s.sh_degree; A.W = s.image_width; A.H = s.image_height;
    A.tanfovx = s.tanfovx; A.tanfovy = s.tanfovy; A.mod = s.scale_modifier; A.depth_mode = s.depth_key; A.tile_bounds = s.tile_bounds;
    A.rec_opacity = (count_into != nullptr && s.tile_bounds != BAGS_TILES_OPACITY) ? 1 : 0;
    A.means3D = in.means3D; A.means2D = in.means2D; A.shs = in.shs; A.colors_precomp = in.colors_precomp; A.opacities = in.opacities;
    A.scales = in.scales; A.rotations = in.rotations; A.cov3D_precomp = in.cov3D_precomp; A.shjac = g.shjac; A.shs_rest = in.shs_rest;
    K1Outputs O;
    O.depth_key = g.depth_key; O.g2d = g.g2d; O.rect = g.rect; O.tiles_touched = g.tiles_touched; O.keep = g.keep; O.radii = radii;
    O.mean2D = mean2D; O.rec_count = g.rec_count;
    if (count_into) {                                        // tile-binned path: K1 also counts the (block, tile) matrix
        const int gy = cdiv(s.image_height, BAGS_TILE), T = grid_x * gy, T2 = (T + 1) / 2;
        const int per = binned_per_block(P), B = cdiv(P, per);
        const size_t lds = (size_t)T2 * 4;
        const void* fn = in.shs_rest ? reinterpret_cast<const void*>(preprocess_fwd_count_kernel<true>)
                                     : reinterpret_cast<const void*>(preprocess_fwd_count_kernel<false>);
        if (lds + 1024 > 65536) {                            // beyond the default 64 KB of LDS per workgroup the launch has to opt in
            hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
#define LAUNCH_COUNT(SP)                                                                                                       \
        LAUNCH_K(preprocess_fwd_count_kernel<SP>, dim3(B), dim3(BIN_THREADS), lds, st, A, s.viewmatrix, s.projmatrix,  \
                           s.intrinsic, s.campos, in.shift_factors, O, per, grid_x, T2, count_into->cnt_rows, g.local_off, g.block_total)
        if (in.shs_rest) LAUNCH_COUNT(true); else LAUNCH_COUNT(false);
#undef LAUNCH_COUNT
        return hipGetLastError();
    }
    if (in.shs_rest)
        LAUNCH_K(preprocess_fwd_kernel<true>, dim3(cdiv(P, 256)), dim3(256), 0, st, A, s.viewmatrix, s.projmatrix, s.intrinsic,
                           s.campos, in.shift_factors, O);
    else
        LAUNCH_K(preprocess_fwd_kernel<false>, dim3(cdiv(P, 256)), dim3(256), 0, st, A, s.viewmatrix, s.projmatrix, s.intrinsic,
                           s.campos, in.shift_factors, O);
    return hipGetLastError();
}
