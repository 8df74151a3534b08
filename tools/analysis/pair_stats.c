/*
 * pair_stats.c — offline (CPU) statistics of the compositing BACKWARD's work on a projected scene: for several shapes of the
 * culling unit (the block of pixels that is walked together for one splat), how many (unit, splat) candidates a kernel
 * would visit and what share of their pixel slots is valid.  Development tool: not product, not oracle.
 *
 * Valid pair = pixel p takes splat at list index i: i < last[p] (the forward's last contributor), sigma >= 0,
 * min(alpha_max, o e^-sigma) >= 1/255 — the rule of composite_oracle.c, in fp32-ish doubles (statistics, not parity).
 *
 * A unit of shape (uw x uh) is a candidate for list entry i of its tile when
 *   geometric: some pixel of the unit has alpha >= 1/255 (what an exact ellipse-vs-box test approaches), and
 *   depth:     i < max over the unit's pixels of last[p].
 * "ideal" candidates additionally need a VALID pixel (no kernel can know that without evaluating).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define TILE 16
#define NSHAPES 10
static const int SHAPES[NSHAPES][2] = {{16, 16}, {16, 8}, {8, 8}, {16, 4}, {8, 4}, {4, 4}, {16, 2}, {8, 2}, {4, 2}, {16, 1}};

/* out[shape][0..5]: geometric&depth candidates, ideal candidates, valid pairs (same for all), candidates geometric only,
 * sum over candidates of (max valid pixel count restricted...) unused, units */
void pair_stats(int inria, int64_t n_isects, const float* means2d, const float* conics, const float* opacities,
                int width, int height, int tile_w, int tile_h, const int32_t* offsets, const int32_t* flatten_ids,
                double* out /* [NSHAPES][6] */, double* hist_valid /* [257]: valid pixels per processed tile entry */,
                double* layouts /* [8]: sum over waves of the longest unit queue, see below */,
                double* rounds_model /* [4][3]: round length 32 / 64 / 128 / whole list -> (rounds, iterations, staged entries) */) {
    const double alpha_max = inria ? (double)0.99f : (double)0.999f;
    const double centre = inria ? 0.0 : 0.5;
    const double amin = (double)(1.0f / 255.0f), tstop = (double)1e-4f;
    const int n_tiles = tile_w * tile_h;
    memset(out, 0, sizeof(double) * NSHAPES * 6);
    memset(hist_valid, 0, sizeof(double) * 257);
    memset(layouts, 0, sizeof(double) * 8);
    memset(rounds_model, 0, sizeof(double) * 12);
#pragma omp parallel
    {
        double loc[NSHAPES][6];
        double lhist[257];
        double llay[8];
        double lrm[12];
        memset(lrm, 0, sizeof(lrm));
        memset(llay, 0, sizeof(llay));
        memset(loc, 0, sizeof(loc));
        memset(lhist, 0, sizeof(lhist));
#pragma omp for schedule(dynamic, 4)
        for (int tile = 0; tile < n_tiles; ++tile) {
            const int start = offsets[tile];
            const int end = (tile + 1 < n_tiles) ? offsets[tile + 1] : (int)n_isects;
            const int tx = tile % tile_w, ty = tile / tile_w;
            const int len = end - start;
            if (len <= 0) continue;
            /* forward per pixel: last index + per (entry) bitmaps */
            int last[256];
            uint8_t* reach = (uint8_t*)malloc((size_t)len * 256);   /* 1 = alpha>=1/255 geometrically, 2 = valid */
            memset(reach, 0, (size_t)len * 256);
            for (int p = 0; p < 256; ++p) {
                const int px = tx * TILE + (p & 15), py = ty * TILE + (p >> 4);
                last[p] = 0;
                if (px >= width || py >= height) continue;
                const double pxf = px + centre, pyf = py + centre;
                double T = 1.0;
                int done = 0;
                for (int i = 0; i < len; ++i) {
                    const int g = flatten_ids[start + i];
                    const double dx = (double)means2d[g * 2] - pxf, dy = (double)means2d[g * 2 + 1] - pyf;
                    const double sigma = 0.5 * ((double)conics[g * 3] * dx * dx + (double)conics[g * 3 + 2] * dy * dy) + (double)conics[g * 3 + 1] * dx * dy;
                    const double raw = (double)opacities[g] * exp(-sigma);
                    const double alpha = raw < alpha_max ? raw : alpha_max;
                    if (sigma < 0.0 || alpha < amin) continue;
                    reach[(size_t)i * 256 + p] |= 1;
                    if (done) continue;
                    const double nT = T * (1.0 - alpha);
                    if (inria ? (nT < tstop) : (nT <= tstop)) { done = 1; continue; }
                    T = nT;
                    last[p] = i + 1;
                    reach[(size_t)i * 256 + p] |= 2;
                }
            }
            int tile_last = 0;
            for (int p = 0; p < 256; ++p) if (last[p] > tile_last) tile_last = last[p];
            for (int i = 0; i < tile_last; ++i) {
                int nv = 0;
                for (int p = 0; p < 256; ++p) nv += (reach[(size_t)i * 256 + p] & 2) ? 1 : 0;
                lhist[nv] += 1.0;
            }
            for (int s = 0; s < NSHAPES; ++s) {
                const int uw = SHAPES[s][0], uh = SHAPES[s][1];
                for (int uy = 0; uy < TILE; uy += uh) for (int ux = 0; ux < TILE; ux += uw) {
                    int ulast = 0;
                    for (int y = uy; y < uy + uh; ++y) for (int x = ux; x < ux + uw; ++x) if (last[y * 16 + x] > ulast) ulast = last[y * 16 + x];
                    loc[s][5] += 1.0;
                    for (int i = 0; i < len; ++i) {
                        int geo = 0, nval = 0;
                        for (int y = uy; y < uy + uh; ++y) for (int x = ux; x < ux + uw; ++x) {
                            const uint8_t r = reach[(size_t)i * 256 + y * 16 + x];
                            geo |= r & 1;
                            nval += (r >> 1) & 1;
                        }
                        if (geo) loc[s][3] += 1.0;
                        if (geo && i < ulast) loc[s][0] += 1.0;
                        if (nval) loc[s][1] += 1.0;
                        loc[s][2] += nval;
                    }
                }
            }
            /* per-unit queue lengths (geometric & depth candidates) on a 4x4-cell grid, combined into the layouts:
             * [0] wave 16x8, one queue (today)           [1] wave 16x8, two 8x8 queues (max)
             * [2] wave 16x8, four 8x4 queues (2x2)       [3] wave 16x8, four 16x2 strips
             * [4] wave 8x8 (1 px/lane), four 4x4 queues  [5] wave 16x8, eight 4x4 queues
             * [6] wave 16x16, four 8x8 queues            [7] wave 16x16, eight 8x4 queues */
            {
                int q16x8[2] = {0, 0}, q8x8[4] = {0}, q8x4[8] = {0}, q16x2[8] = {0}, q4x4[16] = {0};
                int l16x8[2] = {0, 0}, l8x8[4] = {0}, l8x4[8] = {0}, l16x2[8] = {0}, l4x4[16] = {0};
                for (int p = 0; p < 256; ++p) {
                    const int x = p & 15, y = p >> 4, L = last[p];
                    if (L > l16x8[y >> 3]) l16x8[y >> 3] = L;
                    if (L > l8x8[(y >> 3) * 2 + (x >> 3)]) l8x8[(y >> 3) * 2 + (x >> 3)] = L;
                    if (L > l8x4[(y >> 2) * 2 + (x >> 3)]) l8x4[(y >> 2) * 2 + (x >> 3)] = L;
                    if (L > l16x2[y >> 1]) l16x2[y >> 1] = L;
                    if (L > l4x4[(y >> 2) * 4 + (x >> 2)]) l4x4[(y >> 2) * 4 + (x >> 2)] = L;
                }
                for (int i = 0; i < len; ++i) {
                    unsigned g16x8 = 0, g8x8 = 0, g8x4 = 0, g16x2 = 0, g4x4 = 0;
                    for (int p = 0; p < 256; ++p) if (reach[(size_t)i * 256 + p] & 1) {
                        const int x = p & 15, y = p >> 4;
                        g16x8 |= 1u << (y >> 3); g8x8 |= 1u << ((y >> 3) * 2 + (x >> 3)); g8x4 |= 1u << ((y >> 2) * 2 + (x >> 3));
                        g16x2 |= 1u << (y >> 1); g4x4 |= 1u << ((y >> 2) * 4 + (x >> 2));
                    }
                    for (int u = 0; u < 2; ++u) if (((g16x8 >> u) & 1) && i < l16x8[u]) q16x8[u]++;
                    for (int u = 0; u < 4; ++u) if (((g8x8 >> u) & 1) && i < l8x8[u]) q8x8[u]++;
                    for (int u = 0; u < 8; ++u) if (((g8x4 >> u) & 1) && i < l8x4[u]) q8x4[u]++;
                    for (int u = 0; u < 8; ++u) if (((g16x2 >> u) & 1) && i < l16x2[u]) q16x2[u]++;
                    for (int u = 0; u < 16; ++u) if (((g4x4 >> u) & 1) && i < l4x4[u]) q4x4[u]++;
                }
#define MAX2(a, b) ((a) > (b) ? (a) : (b))
                for (int h = 0; h < 2; ++h) {
                    llay[0] += q16x8[h];
                    llay[1] += MAX2(q8x8[2 * h], q8x8[2 * h + 1]);
                    llay[2] += MAX2(MAX2(q8x4[4 * h], q8x4[4 * h + 1]), MAX2(q8x4[4 * h + 2], q8x4[4 * h + 3]));
                    llay[3] += MAX2(MAX2(q16x2[4 * h], q16x2[4 * h + 1]), MAX2(q16x2[4 * h + 2], q16x2[4 * h + 3]));
                    int m = 0;
                    for (int u = 0; u < 8; ++u) m = MAX2(m, q4x4[8 * h + u]);
                    llay[5] += m;
                }
                for (int q = 0; q < 4; ++q) {     /* quadrant q = (qy, qx): its 4x4 cells */
                    const int qy = q >> 1, qx = q & 1;
                    int m = 0;
                    for (int cy = 0; cy < 2; ++cy) for (int cx = 0; cx < 2; ++cx) m = MAX2(m, q4x4[(qy * 2 + cy) * 4 + qx * 2 + cx]);
                    llay[4] += m;
                }
                { int m = 0; for (int u = 0; u < 4; ++u) m = MAX2(m, q8x8[u]); llay[6] += m; }
                { int m = 0; for (int u = 0; u < 8; ++u) m = MAX2(m, q8x4[u]); llay[7] += m; }
            }
            /* composite_bwd4_kernel as built: eight 8x4 units, rounds of R list entries counted back from the tile's deepest last
             * contributor, a round runs max_u (queue length of unit u in this round) iterations */
            {
                int l8x4[8] = {0};
                for (int p = 0; p < 256; ++p) { const int x = p & 15, y = p >> 4; if (last[p] > l8x4[(y >> 2) * 2 + (x >> 3)]) l8x4[(y >> 2) * 2 + (x >> 3)] = last[p]; }
                const int RL[4] = {32, 64, 128, 1 << 30};
                for (int r = 0; r < 4; ++r) {
                    for (int hi = tile_last; hi > 0; hi -= RL[r]) {
                        const int lo = hi - RL[r] > 0 ? hi - RL[r] : 0;
                        int q[8] = {0};
                        for (int i = lo; i < hi; ++i) {
                            unsigned g8 = 0;
                            for (int p = 0; p < 256; ++p) if (reach[(size_t)i * 256 + p] & 1) g8 |= 1u << (((p >> 4) >> 2) * 2 + ((p & 15) >> 3));
                            for (int u = 0; u < 8; ++u) if (((g8 >> u) & 1) && i < l8x4[u]) q[u]++;
                        }
                        int m = 0;
                        for (int u = 0; u < 8; ++u) if (q[u] > m) m = q[u];
                        lrm[r * 3 + 0] += 1.0; lrm[r * 3 + 1] += m; lrm[r * 3 + 2] += hi - lo;
                        if (RL[r] == (1 << 30)) break;
                    }
                }
            }
            free(reach);
        }
#pragma omp critical
        {
            for (int s = 0; s < NSHAPES; ++s) for (int k = 0; k < 6; ++k) out[s * 6 + k] += loc[s][k];
            for (int k = 0; k < 257; ++k) hist_valid[k] += lhist[k];
            for (int k = 0; k < 8; ++k) layouts[k] += llay[k];
            for (int k = 0; k < 12; ++k) rounds_model[k] += lrm[k];
        }
    }
}
