// lincomb_mask_loss (reference modules/yolact.py:241-291) forward AND backward on the f32 MFMA pipe, blockIdx.y = image.
//
//   loss_i = sum_p  w_p * sum_pix BCE( crop_p( sigmoid(proto[pix] . coef[p]) ), gt_p[pix] )      w_p = scale / area_p
//   dproto[pix][k] = sum_p g[p][pix] * coef[p][k]        dcoef[p][k] = sum_pix g[p][pix] * proto[pix][k]
//   g = dL/dz = inside * (m - t) / max(m(1-m), 1e-12) * s(1-s) * w_p * gscale        (torch's BCE backward, m = inside*s)
//
// One wave per 32-pixel tile (grid-stride).  Z^T = coef x proto^T gives D[i = p][j = pix]: lane = pixel, the 16 registers
// = 16 positives -> loss terms + G in registers, and (as in the attention kernel) G is already the A operand of
// dproto = G^T-layout x coef.  For dcoef the reduction runs over pixels, so the same operand registers are multiplied the
// other way round (Z = proto x coef^T: lane = positive, registers = pixels), G' feeds dcoef[i = p][j = k] += G' x proto,
// accumulated in registers over all of the wave's pixel tiles; per-wave partials go to a workspace and are summed in
// fixed order (deterministic, no float atomics).  K = 32 lives in registers: no LDS except 3 KB of per-positive
// metadata.  Algorithmic bytes: P*32*4 (proto) + P*32*4 (dproto) + n_gt*P*4 (gt masks): HBM/L2-bound, ~5 MB per image.
#include "ym_common.h"

namespace {

constexpr int MAXP = 128;

constexpr int MLB = 16;                 // images per launch (per-image arguments travel by value)

struct MLItem {
    const float* proto;      // [P][32]
    const float* coef;       // rows of 32 coefficients: row q of the positives, or row rows[q] of the full [N][32] tensor
    const float* boxes;      // [.][4]  matched gt boxes (crop window + area), indexed like coef
    const int* gt_idx32;     // [n] which downsampled gt mask each positive is trained against (gathered form), or null:
    const int64_t* gt_idx64; //     [N] int64, indexed through rows
    const int64_t* rows;     // [n] anchor index of each positive (null: coef / boxes are already gathered)
    const float* dsmask;     // [n_gt][P]  {0,1}
    float* dproto;           // [P][32]
    float* part;             // [nwaves][MAXP][32] per-wave dcoef partials
    const int* n_dev;        // null, or the device-side count of positives: n_eff = min(*n_dev, n), wscale = *n_dev / n_eff
    int n;
    float wscale;            // old_num_pos / num_pos (sub-sampling correction) — multiplies 1/area
};

struct MLP {
    MLItem it[MLB];
    double* loss;            // accumulated (atomicAdd)
    const int* total_dev;    // null, or the device-side total positive count: gscale /= *total_dev
    int Hp, Wp, P, ntiles;
    float gscale;            // d(total loss)/d(loss_i) = mask_alpha / Hp / Wp / total_pos
};

__device__ __forceinline__ void crop_span(float a, float b, float size, float& lo, float& hi) {
    a = a * size; b = b * size;
    lo = fminf(a, b); hi = fmaxf(a, b);
    lo = lo - 1.f; lo = lo < 0.f ? 0.f : lo;
    hi = hi + 1.f; hi = hi > size ? size : hi;
}

__global__ __launch_bounds__(256) void k_mask_loss(const MLP pb) {
    __shared__ float s_win[MAXP][4];     // x1, x2, y1, y2
    __shared__ float s_w[MAXP];          // wscale / area
    __shared__ int s_gt[MAXP];
    __shared__ int s_row[MAXP];          // coefficient row of each positive
    struct {                             // this image's view, same field names as before
        const float *proto, *coef, *dsmask; float *dproto, *part; double* loss; int n, Hp, Wp, P, ntiles; float wscale, gscale;
    } p;
    const MLItem& it = pb.it[blockIdx.y];
    p.proto = it.proto; p.coef = it.coef; p.dsmask = it.dsmask; p.dproto = it.dproto; p.part = it.part; p.loss = pb.loss;
    p.n = it.n; p.Hp = pb.Hp; p.Wp = pb.Wp; p.P = pb.P; p.ntiles = pb.ntiles; p.wscale = it.wscale; p.gscale = pb.gscale;
    if (it.n_dev) {                      // counts that never left the device (no host synchronisation in the step)
        const int real = *it.n_dev;
        p.n = real < it.n ? real : it.n;
        p.wscale = p.n > 0 ? (float)real / (float)p.n : 0.f;
    }
    if (pb.total_dev) p.gscale = p.gscale / (float)(*pb.total_dev);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int i = tid; i < MAXP; i += 256) {
        if (i < p.n) {
            const int src = it.rows ? (int)it.rows[i] : i;
            s_row[i] = src;
            const f32x4 b = *reinterpret_cast<const f32x4*>(it.boxes + (size_t)src * 4);
            float x1, x2, y1, y2;
            crop_span(b[0], b[2], (float)p.Wp, x1, x2);
            crop_span(b[1], b[3], (float)p.Hp, y1, y2);
            s_win[i][0] = x1; s_win[i][1] = x2; s_win[i][2] = y1; s_win[i][3] = y2;
            s_w[i] = p.wscale / ((b[2] - b[0]) * (b[3] - b[1]));
            s_gt[i] = it.gt_idx32 ? it.gt_idx32[i] : (int)it.gt_idx64[src];
        } else {
            s_row[i] = 0;
            s_win[i][0] = 1.f; s_win[i][1] = 0.f; s_win[i][2] = 1.f; s_win[i][3] = 0.f;   // empty window
            s_w[i] = 0.f; s_gt[i] = 0;
        }
    }
    __syncthreads();
    const int row = lane & 31, h = lane >> 5;
    const int nptile = (p.n + 31) / 32;
    const int wave_id = blockIdx.x * 4 + wv, nwaves = gridDim.x * 4;
    if (p.n == 0) return;                                             // image without positives (uniform)

    f32x16 dc[4];                       // dcoef partial [p tile][D layout: i = p, j = k]
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) dc[t][r] = 0.f;
    double loss_acc = 0.0;

    for (int tile = wave_id; tile < p.ntiles; tile += nwaves) {
        const int pix = tile * 32 + row;
        const bool pix_ok = pix < p.P;
        f32x4 pf[4];
#pragma unroll
        for (int g = 0; g < 4; ++g)
            pf[g] = pix_ok ? *reinterpret_cast<const f32x4*>(p.proto + (size_t)pix * 32 + g * 8 + h * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        const int py = pix / p.Wp, px = pix - py * p.Wp;
        const float fx = (float)px, fy = (float)py;
        f32x16 dp;                       // dproto tile: D[i = pix][j = k]
#pragma unroll
        for (int r = 0; r < 16; ++r) dp[r] = 0.f;

#pragma unroll
        for (int pt = 0; pt < 4; ++pt) {
            if (pt >= nptile) break;                                  // wave-uniform
            const int pp = pt * 32 + row;                              // this lane's positive (operand row / orientation-2 column)
            f32x4 cf[4];
#pragma unroll
            for (int g = 0; g < 4; ++g)
                cf[g] = pp < p.n ? *reinterpret_cast<const f32x4*>(p.coef + (size_t)s_row[pp] * 32 + g * 8 + h * 4) : f32x4{0.f, 0.f, 0.f, 0.f};

            // ---- orientation 1: Z^T[i = positive][j = pixel]: loss + G (lane = pixel) -> dproto ------------------------
            f32x16 zt;
#pragma unroll
            for (int r = 0; r < 16; ++r) zt[r] = 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int s = 0; s < 4; ++s) zt = __builtin_amdgcn_mfma_f32_32x32x2f32(cf[g][s], pf[g][s], zt, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int q = pt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;   // positive index of this register
                float gval = 0.f;
                if (q < p.n && pix_ok) {
                    const float sg = 1.f / (1.f + expf(-zt[r]));
                    const bool inside = fx >= s_win[q][0] && fx < s_win[q][1] && fy >= s_win[q][2] && fy < s_win[q][3];
                    const float m = inside ? sg : 0.f;
                    const float t = p.dsmask[(size_t)s_gt[q] * p.P + pix];
                    const float lm = fmaxf(logf(m), -100.f), l1m = fmaxf(logf(1.f - m), -100.f);
                    loss_acc += (double)(-(t * lm + (1.f - t) * l1m) * s_w[q]);
                    if (inside) gval = (m - t) / fmaxf(m * (1.f - m), 1e-12f) * (sg * (1.f - sg)) * s_w[q] * p.gscale;
                }
                zt[r] = gval;
            }
            // dproto[pix][k] += sum_q G[q][pix] * coef[q][k]: register r of zt is A[i = pix][k-pair member = positive (r, h)]
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int q = pt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const float cq = q < p.n ? p.coef[(size_t)s_row[q] * 32 + row] : 0.f;
                dp = __builtin_amdgcn_mfma_f32_32x32x2f32(zt[r], cq, dp, 0, 0, 0);
            }

            // ---- orientation 2: Z[i = pixel][j = positive]: G' (lane = positive) -> dcoef --------------------------------
            f32x16 z;
#pragma unroll
            for (int r = 0; r < 16; ++r) z[r] = 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int s = 0; s < 4; ++s) z = __builtin_amdgcn_mfma_f32_32x32x2f32(pf[g][s], cf[g][s], z, 0, 0, 0);
            const float w_q = pp < p.n ? s_w[pp] : 0.f;
            const float wx1 = s_win[pp & (MAXP - 1)][0], wx2 = s_win[pp & (MAXP - 1)][1];
            const float wy1 = s_win[pp & (MAXP - 1)][2], wy2 = s_win[pp & (MAXP - 1)][3];
            const size_t gt_base = (size_t)s_gt[pp & (MAXP - 1)] * p.P;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int pix2 = tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                float gval = 0.f;
                if (pp < p.n && pix2 < p.P) {
                    const int y2 = pix2 / p.Wp, x2 = pix2 - y2 * p.Wp;
                    const bool inside = (float)x2 >= wx1 && (float)x2 < wx2 && (float)y2 >= wy1 && (float)y2 < wy2;
                    if (inside) {
                        const float sg = 1.f / (1.f + expf(-z[r]));
                        const float t = p.dsmask[gt_base + pix2];
                        gval = (sg - t) / fmaxf(sg * (1.f - sg), 1e-12f) * (sg * (1.f - sg)) * w_q * p.gscale;
                    }
                }
                z[r] = gval;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int pix2 = tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const float pv = pix2 < p.P ? p.proto[(size_t)pix2 * 32 + row] : 0.f;
                dc[pt] = __builtin_amdgcn_mfma_f32_32x32x2f32(z[r], pv, dc[pt], 0, 0, 0);
            }
        }
        // dproto tile: D[i = pix][j = k]: col = lane & 31 = k, rows = pixels
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int pix2 = tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (pix2 < p.P) p.dproto[(size_t)pix2 * 32 + row] = dp[r];
        }
    }
    // per-wave dcoef partial: D[i = positive][j = k]
    float* part = p.part + (size_t)wave_id * MAXP * 32;
#pragma unroll
    for (int pt = 0; pt < 4; ++pt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int q = pt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            part[(size_t)q * 32 + row] = dc[pt][r];
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) loss_acc += __shfl_xor(loss_acc, o);
    if (lane == 0 && loss_acc != 0.0) atomicAdd(p.loss, loss_acc);
}

// dcoef_full[anchor_idx[q]][k] = sum over the waves' partials in a fixed order: workgroup = (positive q, image), thread = (slice of
// the waves, k); 8 slices are combined through LDS
struct MLReduce { const float* part[MLB]; const int64_t* anchor_idx[MLB]; float* dcoef_full[MLB]; const int* n_dev[MLB]; int n[MLB]; };

__global__ __launch_bounds__(256) void k_mask_loss_reduce(const MLReduce rb, int nwaves) {
    __shared__ float s[8][32];
    const int img = blockIdx.y, q = blockIdx.x;
    int n = rb.n[img];
    if (rb.n_dev[img]) n = min(n, *rb.n_dev[img]);
    if (q >= n) return;
    const int k = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const float* part = rb.part[img];
    const int per = (nwaves + 7) / 8;
    float v = 0.f;
    for (int w = sl * per; w < min(nwaves, (sl + 1) * per); ++w) v += part[((size_t)w * MAXP + q) * 32 + k];
    s[sl][k] = v;
    __syncthreads();
    if (sl == 0) {
        float t = s[0][k];
#pragma unroll
        for (int i = 1; i < 8; ++i) t += s[i][k];
        rb.dcoef_full[img][(size_t)rb.anchor_idx[img][q] * 32 + k] = t;
    }
}

constexpr int ML_BLOCKS = 64;    // 256 waves per image, each walking ~P/32/256 pixel tiles
constexpr size_t ML_PART_BYTES = (size_t)ML_BLOCKS * 4 * MAXP * 32 * sizeof(float);

int launch_chunk(const ym_mask_loss_item* items, int nb, const float* const* coef, const float* const* boxes, const int* const* gt32,
                 int Hp, int Wp, float gscale, const int32_t* total_dev, double* loss_accum, char* workspace, hipStream_t st) {
    MLP p;
    MLReduce r;
    int nmax = 0;
    for (int i = 0; i < nb; ++i) {
        const ym_mask_loss_item& a = items[i];
        MLItem& it = p.it[i];
        it.proto = a.proto; it.dsmask = a.gt_masks_ds; it.dproto = a.dproto; it.n = a.n; it.wscale = a.wscale; it.n_dev = a.n_dev;
        it.part = (float*)(workspace + (size_t)i * ML_PART_BYTES);
        if (coef) { it.coef = coef[i]; it.boxes = boxes[i]; it.gt_idx32 = gt32[i]; it.gt_idx64 = nullptr; it.rows = nullptr; }
        else { it.coef = a.coef_full; it.boxes = a.anchor_box; it.gt_idx32 = nullptr; it.gt_idx64 = a.anchor_gt; it.rows = a.anchor_idx; }
        r.part[i] = it.part; r.anchor_idx[i] = a.anchor_idx; r.dcoef_full[i] = a.dcoef_full; r.n[i] = a.n; r.n_dev[i] = a.n_dev;
        if (a.n > nmax) nmax = a.n;
    }
    if (nmax == 0) return YM_OK;
    p.loss = loss_accum; p.total_dev = total_dev; p.Hp = Hp; p.Wp = Wp; p.P = Hp * Wp; p.ntiles = (p.P + 31) / 32; p.gscale = gscale;
    hipLaunchKernelGGL(k_mask_loss, dim3(ML_BLOCKS, nb), dim3(256), 0, st, p);
    hipLaunchKernelGGL(k_mask_loss_reduce, dim3(nmax, nb), dim3(256), 0, st, r, ML_BLOCKS * 4);
    return ym_check_launch("mask_loss");
}

}  // namespace

extern "C" size_t ym_mask_loss_workspace_bytes(void) { return ML_PART_BYTES + 256; }
extern "C" size_t ym_mask_loss_batch_workspace_bytes(int B) { return (size_t)(B < MLB ? (B > 0 ? B : 1) : MLB) * ML_PART_BYTES + 256; }

extern "C" int ym_mask_loss_batch(const ym_mask_loss_item* items, int B, int Hp, int Wp, float gscale, const int32_t* total_pos_dev,
                                  double* loss_accum, void* workspace, size_t workspace_bytes, ym_stream_t s) {
    YM_REQUIRE(items && B > 0 && Hp > 0 && Wp > 0 && loss_accum && workspace, "mask_loss_batch: bad args");
    for (int i = 0; i < B; ++i) {
        const ym_mask_loss_item& a = items[i];
        YM_REQUIRE(a.n >= 0 && a.n <= MAXP, "mask_loss: at most %d positives per image (cfg.masks_to_train), got %d", MAXP, a.n);
        YM_REQUIRE(a.n == 0 || (a.proto && a.coef_full && a.anchor_box && a.anchor_gt && a.gt_masks_ds && a.anchor_idx && a.dproto && a.dcoef_full),
                   "mask_loss_batch: null pointer in item %d", i);
    }
    if (workspace_bytes < ym_mask_loss_batch_workspace_bytes(B)) { ym_set_error("mask_loss_batch: workspace too small"); return YM_ENOSPC; }
    for (int b0 = 0; b0 < B; b0 += MLB) {
        const int nb = B - b0 < MLB ? B - b0 : MLB;
        const int rc = launch_chunk(items + b0, nb, nullptr, nullptr, nullptr, Hp, Wp, gscale, total_pos_dev, loss_accum, (char*)workspace, (hipStream_t)s);
        if (rc != YM_OK) return rc;
    }
    return YM_OK;
}

extern "C" int ym_mask_loss_fwd_bwd(const float* proto, const float* coef_pos, const float* box_pos, const int32_t* gt_idx,
                                    const float* gt_masks_ds, const int64_t* anchor_idx, int n, int Hp, int Wp, float wscale,
                                    float gscale, double* loss_accum, float* dproto, float* dcoef_full, void* workspace,
                                    size_t workspace_bytes, ym_stream_t s) {
    YM_REQUIRE(n >= 0 && n <= MAXP, "mask_loss: at most %d positives per image (cfg.masks_to_train), got %d", MAXP, n);
    if (n == 0) return YM_OK;
    YM_REQUIRE(proto && coef_pos && box_pos && gt_idx && gt_masks_ds && anchor_idx && loss_accum && dproto && dcoef_full && workspace,
               "mask_loss: null pointer");
    if (workspace_bytes < ym_mask_loss_workspace_bytes()) { ym_set_error("mask_loss: workspace too small"); return YM_ENOSPC; }
    ym_mask_loss_item a{};
    a.proto = proto; a.gt_masks_ds = gt_masks_ds; a.anchor_idx = anchor_idx; a.n = n; a.wscale = wscale; a.dproto = dproto;
    a.dcoef_full = dcoef_full;
    const int* g32 = gt_idx;
    return launch_chunk(&a, 1, &coef_pos, &box_pos, &g32, Hp, Wp, gscale, nullptr, loss_accum, (char*)workspace, (hipStream_t)s);
}
