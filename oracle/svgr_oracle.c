/*
 * svgr_oracle.c -- CPU restatement (plain C, IEEE double) of the reference hot path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load this library.  The product path
 * (svgrasterize.py_amd/ + libsvgr_hip.so) never links, imports or calls it.
 *
 * Parity status: PINNED.  Every function below is checked bit-for-bit (or to the stated
 * tolerance) against fixtures produced by running the reference itself in the build
 * container (oracle/gen_golden.py -> tests/golden/ *.npz; tests/test_oracle_*.py).
 *
 * "S:n" = /root/reference/svgrasterize.py line n.  The arithmetic forms marked [measured]
 * are what numpy 2.2.6 + scipy-openblas 0.3.29 evaluate on the build container for the
 * reference's np.dot / @ calls (probed against 1e4 random operands, 100 % match):
 *   Transform.__call__ (S:531-534):       out_r = fma(p1, m_r1, p0*m_r0) + b_r
 *   np.dot(W(k,4), batch(N,4,2)) (S:2068, S:2087): fma(w0,x0, w2*x2) + fma(w1,x1, w3*x3)
 * Everything else in the path is plain separately-rounded double arithmetic, so this
 * file must be compiled with -ffp-contract=off.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* ---------------------------------------------------------------------------------- */
/* Transform.__call__  S:531-534                                                      */
/* m6 = {m00, m01, m02, m10, m11, m12}; points are (.., 2) pairs                      */
/* ---------------------------------------------------------------------------------- */
ORC_API void orc_transform_points(const double *m6, const double *in, int64_t npts, double *out)
{
    for (int64_t i = 0; i < npts; ++i) {
        double p0 = in[2 * i], p1 = in[2 * i + 1];
        out[2 * i] = fma(p1, m6[1], p0 * m6[0]) + m6[2];
        out[2 * i + 1] = fma(p1, m6[4], p0 * m6[3]) + m6[5];
    }
}

/* 3x3 affine product as numpy evaluates `a.m @ b.m` (S:523): fma(a2,b2, fma(a1,b1, a0*b0)) */
ORC_API void orc_matmul3(const double *a, const double *b, double *c)
{
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            c[3 * i + j] = fma(a[3 * i + 2], b[6 + j], fma(a[3 * i + 1], b[3 + j], a[3 * i] * b[j]));
}

/* ---------------------------------------------------------------------------------- */
/* Bezier flatten  S:2035-2098                                                        */
/* ---------------------------------------------------------------------------------- */
static inline double dot4(const double w[4], double x0, double x1, double x2, double x3)
{
    /* [measured] strided ddot form */
    return fma(w[0], x0, w[2] * x2) + fma(w[1], x1, w[3] * x3);
}

static const double FLATNESS_W[2][4] = {{-2, 3, 0, -1}, {-1, 0, 3, -2}}; /* S:2035 */
static const double SPLIT_W[8][4] = {                                    /* S:2036-2048 */
    {1, 0, 0, 0},          {0.5, 0.5, 0, 0},   {0.25, 0.5, 0.25, 0}, {0.125, 0.375, 0.375, 0.125},
    {0.125, 0.375, 0.375, 0.125}, {0, 0.25, 0.5, 0.25}, {0, 0, 0.5, 0.5},     {0, 0, 0, 1}};

/* bezier3_flatness_batch S:2071-2088.  NOTE: the code (not its docstring) reduces over the
 * COORDINATE axis first: uv has shape (N, coord, {u,v}); .max(-2) is over coord, .sum(-1) over
 * {u, v}:   f = max(ux^2, uy^2) + max(vx^2, vy^2)                                        */
static inline double cubic_flatness(const double *c /* 4x2 */)
{
    double f = 0.0;
    for (int k = 0; k < 2; ++k) { /* k = 0: u, k = 1: v */
        double tx = dot4(FLATNESS_W[k], c[0], c[2], c[4], c[6]);
        double ty = dot4(FLATNESS_W[k], c[1], c[3], c[5], c[7]);
        double xx = tx * tx, yy = ty * ty; /* np.square */
        double mx = xx > yy ? xx : yy;
        f = (k == 0) ? mx : f + mx;
    }
    return f;
}

/* bezier3_split_batch S:2066-2068: out = [left(4x2), right(4x2)] */
static inline void cubic_split(const double *c, double *out /* 8x2 */)
{
    for (int r = 0; r < 8; ++r)
        for (int ax = 0; ax < 2; ++ax)
            out[2 * r + ax] = dot4(SPLIT_W[r], c[ax], c[2 + ax], c[4 + ax], c[6 + ax]);
}

ORC_API void orc_flatness(const double *cubics, int64_t n, double *out)
{
    for (int64_t i = 0; i < n; ++i) out[i] = cubic_flatness(cubics + 8 * i);
}

ORC_API void orc_split(const double *cubics, int64_t n, double *out)
{
    for (int64_t i = 0; i < n; ++i) cubic_split(cubics + 8 * i, out + 16 * i);
}

/*
 * bezier3_flatten_batch S:2091-2098, level-synchronous, same emission order as the
 * reference (all flat curves of level 0, then level 1, ...).  Threshold is
 * (flatness**2)*16 evaluated in double exactly as Python does (0.1 -> 0.16000000000000003).
 * Returns the number of edges; if `edges` is NULL only counts.  `cap` = capacity of `edges`
 * in edges (4 doubles each); returns -1 if it would overflow, -2 on allocation failure,
 * -3 if a level limit of 64 is hit (non-finite input would loop forever in the reference).
 */
ORC_API int64_t orc_flatten(const double *cubics, int64_t n, double flatness, double *edges, int64_t cap)
{
    double thr = (flatness * flatness) * 16.0;
    int64_t count = 0;
    if (n == 0) return 0;
    double *cur = (double *)malloc(sizeof(double) * 8 * (size_t)n);
    if (!cur) return -2;
    memcpy(cur, cubics, sizeof(double) * 8 * (size_t)n);
    int64_t ncur = n;
    for (int level = 0; ncur > 0; ++level) {
        if (level >= 64) { free(cur); return -3; }
        int64_t nkeep = 0;
        for (int64_t i = 0; i < ncur; ++i)
            if (!(cubic_flatness(cur + 8 * i) < thr)) ++nkeep;
        double *next = nkeep ? (double *)malloc(sizeof(double) * 16 * (size_t)nkeep) : NULL;
        if (nkeep && !next) { free(cur); return -2; }
        int64_t k = 0;
        for (int64_t i = 0; i < ncur; ++i) {
            const double *c = cur + 8 * i;
            if (cubic_flatness(c) < thr) {
                if (edges) {
                    if (count >= cap) { free(cur); free(next); return -1; }
                    double *e = edges + 4 * count;
                    e[0] = c[0]; e[1] = c[1]; e[2] = c[6]; e[3] = c[7];
                }
                ++count;
            } else {
                cubic_split(c, next + 16 * k);
                ++k;
            }
        }
        free(cur);
        cur = next;
        ncur = 2 * nkeep;
    }
    free(cur);
    return count;
}

/* ---------------------------------------------------------------------------------- */
/* bbox  S:966-975.  edges = E x (p0.row, p0.col, p1.row, p1.col).                    */
/* viewport = {r0, c0, rows, cols} or NULL.  out = {min_r, min_c, rows, cols}.        */
/* returns 1 if non-empty                                                             */
/* ---------------------------------------------------------------------------------- */
ORC_API int orc_bbox(const double *edges, int64_t n_edges, const int64_t *viewport, int64_t *out)
{
    if (n_edges == 0) return 0;
    double mn[2] = {INFINITY, INFINITY}, mx[2] = {-INFINITY, -INFINITY};
    for (int64_t i = 0; i < 2 * n_edges; ++i)
        for (int ax = 0; ax < 2; ++ax) {
            double v = edges[2 * i + ax];
            if (v < mn[ax]) mn[ax] = v;
            if (v > mx[ax]) mx[ax] = v;
        }
    int64_t lo[2], hi[2];
    for (int ax = 0; ax < 2; ++ax) {
        lo[ax] = (int64_t)floor(mn[ax]) - 1;
        hi[ax] = (int64_t)ceil(mx[ax]) + 1;
    }
    if (viewport) {
        for (int ax = 0; ax < 2; ++ax) {
            if (viewport[ax] > lo[ax]) lo[ax] = viewport[ax];
            if (viewport[ax] + viewport[2 + ax] < hi[ax]) hi[ax] = viewport[ax] + viewport[2 + ax];
        }
    }
    out[0] = lo[0]; out[1] = lo[1]; out[2] = hi[0] - lo[0]; out[3] = hi[1] - lo[1];
    return out[2] > 0 && out[3] > 0;
}

/* ---------------------------------------------------------------------------------- */
/* line_signed_coverage  S:2213-2304.  trace is (h, w) row-major; line = {r0,c0,r1,c1}*/
/* already relative to the trace origin.  `**2` in the reference is libm pow.          */
/* ---------------------------------------------------------------------------------- */
static inline void acc(double *row, int64_t w, int64_t xi, double v)
{
    row[xi > 0 ? xi : 0] += v; /* left of the canvas folds into column 0 */
    (void)w;
}

ORC_API void orc_line_coverage(double *trace, int64_t h, int64_t w, const double *line)
{
    double p0y = line[0], p0x = line[1], p1y = line[2], p1x = line[3];
    if (p0y == p1y) return;
    double dir = 1.0;
    if (!(p0y < p1y)) {
        dir = -1.0;
        double t = p0y; p0y = p1y; p1y = t;
        t = p0x; p0x = p1x; p1x = t;
    }
    double dxdy = (p1x - p0x) / (p1y - p0y);
    double x = p0x;
    int64_t y = (int64_t)(p0y > 0 ? p0y : 0); /* int(max(0, p0y)) truncates */
    if (p0y < 0) x -= p0y * dxdy;
    double x_next = x;
    int64_t y_end = (int64_t)ceil(p1y);
    if (h < y_end) y_end = h;
    for (; y < y_end; ++y) {
        double *row = trace + y * w;
        x = x_next;
        double yhi = (double)(y + 1) < p1y ? (double)(y + 1) : p1y;
        double ylo = (double)y > p0y ? (double)y : p0y;
        double dy = yhi - ylo;
        double d = dir * dy;
        x_next = x + dxdy * dy;
        double x0 = x < x_next ? x : x_next;
        double x1 = x < x_next ? x_next : x;
        double x0_floor = floor(x0);
        int64_t x0i = (int64_t)x0_floor;
        double x1_ceil = ceil(x1);
        int64_t x1i = (int64_t)x1_ceil;
        if (x1i <= x0i + 1) {
            double xmf = 0.5 * (x + x_next) - x0_floor;
            if (x0i >= w) continue;
            acc(row, w, x0i, d * (1 - xmf));
            if (x0i + 1 >= w) continue;
            acc(row, w, x0i + 1, d * xmf);
        } else {
            double s = 1 / (x1 - x0);
            double x0f = x0 - x0_floor;
            double x1f = x1 - x1_ceil + 1.0;
            double a0 = 0.5 * s * pow(1 - x0f, 2.0);
            double am = 0.5 * s * pow(x1f, 2.0);
            if (x0i >= w) continue;
            acc(row, w, x0i, d * a0);
            if (x1i == x0i + 2) {
                if (x0i + 1 >= w) continue;
                acc(row, w, x0i + 1, d * (1.0 - a0 - am));
            } else {
                double a1 = s * (1.5 - x0f);
                if (x0i + 1 >= w) continue;
                acc(row, w, x0i + 1, d * (a1 - a0));
                int stop = 0;
                for (int64_t xi = x0i + 2; xi < x1i - 1; ++xi) {
                    if (xi >= w) { stop = 1; break; } /* every later xi is also >= w */
                    acc(row, w, xi, d * s);
                }
                (void)stop;
                double a2 = a1 + (double)(x1i - x0i - 3) * s;
                if (x1i - 1 >= w) continue;
                acc(row, w, x1i - 1, d * (1.0 - a2 - am));
            }
            if (x1i >= w) continue;
            acc(row, w, x1i, d * am);
        }
    }
}

/* ---------------------------------------------------------------------------------- */
/* Path.mask core  S:978-990.  origin = {min_r, min_c}; rule 0 = nonzero, 1 = evenodd. */
/* mask is (rows, cols) doubles (zero-initialised here).                              */
/* ---------------------------------------------------------------------------------- */
static inline double py_mod2(double a)
{
    /* np.remainder(a, 2.0): fmod then move into [0, 2) */
    double m = fmod(a, 2.0);
    if (m != 0.0) {
        if (m < 0.0) m += 2.0;
    } else {
        m = 0.0;
    }
    return m;
}

ORC_API void orc_mask(const double *edges, int64_t n_edges, const int64_t *origin, int64_t rows, int64_t cols,
                      int rule, double *mask)
{
    memset(mask, 0, sizeof(double) * (size_t)(rows * cols));
    double o0 = (double)origin[0], o1 = (double)origin[1];
    for (int64_t i = 0; i < n_edges; ++i) {
        double l[4] = {edges[4 * i] - o0, edges[4 * i + 1] - o1, edges[4 * i + 2] - o0, edges[4 * i + 3] - o1};
        orc_line_coverage(mask, rows, cols, l);
    }
    for (int64_t r = 0; r < rows; ++r) {
        double *row = mask + r * cols;
        double s = 0.0;
        for (int64_t c = 0; c < cols; ++c) {
            s = (c == 0) ? row[0] : s + row[c]; /* np.cumsum: first element copied */
            double m;
            if (rule == 0) {
                m = fabs(s);
                if (m > 1.0) m = 1.0; /* clip(0, 1) after fabs */
            } else {
                m = fabs(py_mod2(s + 1.0) - 1.0);
            }
            if (m < 1e-6) m = 0.0;
            row[c] = m;
        }
    }
}

/* ---------------------------------------------------------------------------------- */
/* colour helpers  S:471-503 (on n RGBA quadruples, in place)                          */
/* ---------------------------------------------------------------------------------- */
ORC_API void orc_pre_to_straight(double *rgba, int64_t n)
{
    for (int64_t i = 0; i < n; ++i) {
        double a = rgba[4 * i + 3];
        for (int c = 0; c < 4; ++c) {
            double v = rgba[4 * i + c];
            if (c < 3 && a > 0.0001) v = v / a;
            v = v < 0 ? 0 : (v > 1 ? 1 : v); /* np.clip(rgba, 0, 1) incl. alpha */
            rgba[4 * i + c] = v;
        }
    }
}

ORC_API void orc_straight_to_pre(double *rgba, int64_t n)
{
    for (int64_t i = 0; i < n; ++i)
        for (int c = 0; c < 3; ++c) rgba[4 * i + c] *= rgba[4 * i + 3];
}

ORC_API void orc_linear_to_srgb(double *rgba, int64_t n)
{
    for (int64_t i = 0; i < n; ++i)
        for (int c = 0; c < 3; ++c) {
            double v = rgba[4 * i + c];
            rgba[4 * i + c] = (v <= 0.0031308) ? v * 12.92 : 1.055 * pow(v, 1.0 / 2.4) - 0.055;
        }
}

ORC_API void orc_srgb_to_linear(double *rgba, int64_t n)
{
    for (int64_t i = 0; i < n; ++i)
        for (int c = 0; c < 3; ++c) {
            double v = rgba[4 * i + c];
            rgba[4 * i + c] = (v <= 0.04045) ? v / 12.92 : pow((v + 0.055) / 1.055, 2.4);
        }
}

/* Path.fill solid paint preparation S:1014-1018 */
ORC_API void orc_paint_for_fill(const double *paint, int linear_rgb, double *out)
{
    memcpy(out, paint, 4 * sizeof(double));
    if (!linear_rgb) {
        orc_pre_to_straight(out, 1);
        orc_linear_to_srgb(out, 1);
        orc_straight_to_pre(out, 1);
    }
}

/* image = mask * paint  S:1019 */
ORC_API void orc_fill_solid(const double *mask, int64_t npx, const double *paint, double *rgba)
{
    for (int64_t i = 0; i < npx; ++i)
        for (int c = 0; c < 4; ++c) rgba[4 * i + c] = mask[i] * paint[c];
}

/* ---------------------------------------------------------------------------------- */
/* canvas_compose + merge  S:277-298, 366-377, 382-416                                 */
/* dst: (dr, dc, 4) with origin dst_off; src: (sr, sc, ch) at src_off, ch in {1, 4}.   */
/* Only the overlap is touched (for union merges the overlap is the whole src).        */
/* ---------------------------------------------------------------------------------- */
ORC_API void orc_compose_over(double *dst, const int64_t *dst_off, int64_t dr, int64_t dc, const double *src,
                              const int64_t *src_off, int64_t sr, int64_t sc, int ch, int first)
{
    for (int64_t r = 0; r < sr; ++r) {
        int64_t R = r + src_off[0] - dst_off[0];
        if (R < 0 || R >= dr) continue;
        for (int64_t c = 0; c < sc; ++c) {
            int64_t C = c + src_off[1] - dst_off[1];
            if (C < 0 || C >= dc) continue;
            double *d = dst + 4 * (R * dc + C);
            const double *s = src + ch * (r * sc + c);
            double sa = s[ch - 1];
            for (int k = 0; k < 4; ++k) {
                double sv = ch == 4 ? s[k] : s[0];
                d[k] = first ? sv : sv + d[k] * (1 - sa); /* src + dst * (1 - src_a) */
            }
        }
    }
}

/* COMPOSE_IN step: out = src * dst_alpha, dst is (dr, dc, 4) updated in place over the
 * region both cover (caller guarantees dst == intersection region, S:406-414). */
ORC_API void orc_compose_in(double *dst, const int64_t *dst_off, int64_t dr, int64_t dc, const double *src,
                            const int64_t *src_off, int64_t sr, int64_t sc, int ch)
{
    for (int64_t R = 0; R < dr; ++R) {
        int64_t r = R + dst_off[0] - src_off[0];
        for (int64_t C = 0; C < dc; ++C) {
            int64_t c = C + dst_off[1] - src_off[1];
            double *d = dst + 4 * (R * dc + C);
            double da = d[3];
            if (r < 0 || r >= sr || c < 0 || c >= sc) continue;
            const double *s = src + ch * (r * sc + c);
            for (int k = 0; k < 4; ++k) d[k] = (ch == 4 ? s[k] : s[0]) * da;
        }
    }
}

/* ---------------------------------------------------------------------------------- */
/* Whole solid-fill scene, as Scene.render(GROUP of FILL) + canvas_merge_at does it:   */
/* per path flatten -> bbox -> trace -> cumsum -> rule -> mask*paint -> OVER.          */
/* Used as (a) the large-size parity reference on the GPU box and (b) the cpu_baseline */
/* leg of bench.py.  Per-path work is materialised pass by pass like the reference     */
/* (no fusion); `viewport` lets callers split the canvas into row strips (S:968-971).  */
/*                                                                                     */
/* segs:   n_segs x 8 doubles (4 points; lines use the first two), already in          */
/*         presentation space; seg_kind[i] 0 = line, 1 = cubic                         */
/* path_seg_off: n_paths+1 offsets into segs; path_rule[p]; path_paint: n_paths x 4    */
/*         (premultiplied, already in compositing space)                               */
/* canvas: (vrows, vcols, 4) doubles covering `viewport`, zeroed by the caller.        */
/* stats (optional, 2 x int64): path-pixels P and flattened edges E                    */
/* returns 0 or a negative error                                                       */
/* ---------------------------------------------------------------------------------- */
/* Optional census for bench.py's roofline block: how many path-pixels are visible (coverage != 0 after the cut). */
static int g_count_visible = 0;
static int64_t g_visible = 0;
ORC_API void orc_visible_count(int enable) { g_count_visible = enable; g_visible = 0; }
ORC_API int64_t orc_visible_get(void) { return g_visible; }

ORC_API int orc_render_solid(const double *segs, const uint8_t *seg_kind, const int64_t *path_seg_off, int64_t n_paths,
                             const uint8_t *path_rule, const double *path_paint, const int64_t *viewport,
                             int clip01, double *canvas, int64_t *stats)
{
    int64_t P = 0, E = 0;
    int64_t ecap = 1024;
    double *edges = (double *)malloc(sizeof(double) * 4 * (size_t)ecap);
    double *cub = NULL, *mask = NULL, *rgba = NULL;
    int64_t ccap = 0, pxcap = 0;
    if (!edges) return -2;
    int rc = 0;
    for (int64_t p = 0; p < n_paths && rc == 0; ++p) {
        int64_t s0 = path_seg_off[p], s1 = path_seg_off[p + 1];
        int64_t nl = 0, nc = 0;
        for (int64_t s = s0; s < s1; ++s) (seg_kind[s] ? ++nc : ++nl);
        if (nc > ccap) {
            free(cub);
            ccap = nc * 2;
            cub = (double *)malloc(sizeof(double) * 8 * (size_t)ccap);
            if (!cub) { rc = -2; break; }
        }
        int64_t k = 0;
        for (int64_t s = s0; s < s1; ++s)
            if (seg_kind[s]) memcpy(cub + 8 * k++, segs + 8 * s, 8 * sizeof(double));
        int64_t ne_c = orc_flatten(cub, nc, 0.1, NULL, 0);
        if (ne_c < 0) { rc = (int)ne_c; break; }
        int64_t ne = nl + ne_c;
        if (ne == 0) continue;
        if (ne > ecap) {
            free(edges);
            ecap = ne * 2;
            edges = (double *)malloc(sizeof(double) * 4 * (size_t)ecap);
            if (!edges) { rc = -2; break; }
        }
        k = 0;
        for (int64_t s = s0; s < s1; ++s)
            if (!seg_kind[s]) memcpy(edges + 4 * k++, segs + 8 * s, 4 * sizeof(double));
        orc_flatten(cub, nc, 0.1, edges + 4 * nl, ecap - nl);
        int64_t bb[4];
        if (!orc_bbox(edges, ne, viewport, bb)) continue;
        E += ne;
        P += bb[2] * bb[3];
        int64_t npx = bb[2] * bb[3];
        if (npx > pxcap) { /* grow-only scratch: fresh mmap pages per path would time the page-fault path */
            free(mask);
            free(rgba);
            pxcap = npx + npx / 2;
            mask = (double *)malloc(sizeof(double) * (size_t)pxcap);
            rgba = (double *)malloc(sizeof(double) * 4 * (size_t)pxcap);
            if (!mask || !rgba) { rc = -2; break; }
        }
        orc_mask(edges, ne, bb, bb[2], bb[3], path_rule[p], mask);
        if (g_count_visible) { /* bench.py only: path-pixels whose coverage survives the 1e-6 cut (S:990) */
            int64_t nv = 0;
            for (int64_t i = 0; i < npx; ++i) nv += mask[i] != 0.0;
#pragma omp atomic
            g_visible += nv;
        }
        orc_fill_solid(mask, npx, path_paint + 4 * p, rgba);
        orc_compose_over(canvas, viewport, viewport[2], viewport[3], rgba, bb, bb[2], bb[3], 4, 0);
    }
    free(mask);
    free(rgba);
    if (rc == 0 && clip01) { /* canvas_merge_at(...).clip(0, 1)  S:326 */
        int64_t n = viewport[2] * viewport[3] * 4;
        for (int64_t i = 0; i < n; ++i) canvas[i] = canvas[i] < 0 ? 0 : (canvas[i] > 1 ? 1 : canvas[i]);
    }
    free(edges);
    free(cub);
    if (stats) { stats[0] = P; stats[1] = E; }
    return rc;
}

/* The same render with the canvas cut into horizontal strips, one OpenMP thread per strip at a time: every strip is  */
/* an independent orc_render_solid with its own viewport (the reference's own cropping, S:968-971), so the threads  */
/* share nothing.  Used only for the "all host cores" CPU baseline of bench.py.  stats[0] = P.          */
ORC_API int orc_render_solid_strips(const double *segs, const uint8_t *seg_kind, const int64_t *path_seg_off, int64_t n_paths,
                                    const uint8_t *path_rule, const double *path_paint, const int64_t *viewport,
                                    int clip01, double *canvas, int64_t *stats, int n_strips, int n_threads)
{
    if (n_strips < 1) n_strips = 1;
    int64_t rows = viewport[2], cols = viewport[3];
    int64_t h = (rows + n_strips - 1) / n_strips;
    int rc_all = 0;
    int64_t P = 0;
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads) reduction(+ : P)
    for (int s = 0; s < n_strips; ++s) {
        int64_t r0 = s * h, r1 = r0 + h < rows ? r0 + h : rows;
        if (r0 >= r1) continue;
        int64_t vp[4] = {viewport[0] + r0, viewport[1], r1 - r0, cols};
        int64_t st[2] = {0, 0};
        int rc = orc_render_solid(segs, seg_kind, path_seg_off, n_paths, path_rule, path_paint, vp, clip01,
                                  canvas + (size_t)r0 * (size_t)cols * 4, st);
        if (rc != 0) {
#pragma omp critical
            rc_all = rc;
        }
        P += st[0];
    }
    if (stats) { stats[0] = P; stats[1] = 0; }
    return rc_all;
}
