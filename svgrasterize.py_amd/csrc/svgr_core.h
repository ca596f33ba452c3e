// svgr_core.h -- per-lane arithmetic of the hot path, shared by every HIP kernel.
//
// Everything here is straight-line double arithmetic written so that one lane reproduces
// what the reference evaluates for one element (S:n = svgrasterize.py line n of the reference):
//   xform_point      Transform.__call__            S:531-534
//   cubic_flatness   bezier3_flatness_batch        S:2071-2088
//   cubic_split      bezier3_split_batch           S:2066-2068
//   flatten_cubic    bezier3_flatten_batch         S:2091-2098 (depth-first instead of level-synchronous:
//                                                   same edge SET, curve order instead of level order)
//   EdgeWalk         line_signed_coverage          S:2213-2304 (row recurrence + per-row area pieces)
//   fill_rule_*      Path.mask                     S:984-990
//   over_px          canvas_compose(OVER)          S:286
// The functions are SVGR_HD so the same code can be compiled for the host by the unit-test
// harness (tests/host_harness.cpp); the product only ever runs them inside HIP kernels.
// Compile with -ffp-contract=off: the reference rounds every operation separately except
// where an explicit fma() below mirrors what BLAS evaluates for np.dot / @.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define SVGR_HD __host__ __device__ __forceinline__
#else
#define SVGR_HD static inline
#endif

namespace svgr {

constexpr int kMaxFlattenDepth = 40;   // reference has no cap; finite input never gets close
constexpr double kZeroCut = 1e-6;      // S:990

// ------------------------------------------------------------------------------------
// affine transform of one point: out_r = fma(p1, m_r1, p0*m_r0) + b_r     (dgemm form)
// m6 = {m00, m01, m02, m10, m11, m12}
// ------------------------------------------------------------------------------------
SVGR_HD void xform_point(const double* m6, double p0, double p1, double& o0, double& o1) {
    o0 = fma(p1, m6[1], p0 * m6[0]) + m6[2];
    o1 = fma(p1, m6[4], p0 * m6[3]) + m6[5];
}

// strided-ddot form of np.dot(W(k,4), batch(N,4,2)): fma(w0,x0, w2*x2) + fma(w1,x1, w3*x3)
SVGR_HD double dot4(double w0, double w1, double w2, double w3, double x0, double x1, double x2, double x3) {
    return fma(w0, x0, w2 * x2) + fma(w1, x1, w3 * x3);
}

// The weight rows used below contain zeros and ones.  With w == 0 the term fma(0, x, t) is t and 0 * x is +0, with
// w == 1 fma(1, x, t) is x + t: for finite x the shortened forms give the same bits as the full dot4 (up to the sign
// of a zero, which nothing downstream can see), at about half the instructions -- the compiler may not drop them
// itself (0 * x is not 0 for a NaN or an infinity).

// c = 4 points (row, col) interleaved: c[2*k + axis]
SVGR_HD double cubic_flatness(const double* c) {
    // u = -2 b0 + 3 b1 - b3 ; v = -b0 + 3 b2 - 2 b3 ; f = max(ux^2, uy^2) + max(vx^2, vy^2)
    // dot4(-2, 3, 0, -1, x) = fma(-2, x0, 0 * x2) + fma(3, x1, -1 * x3) = -2 x0 + fma(3, x1, -x3)
    // dot4(-1, 0, 3, -2, x) = fma(-1, x0, 3 * x2) + fma(0, x1, -2 * x3) = fma(-1, x0, 3 x2) + -2 x3
    double ux = -2.0 * c[0] + fma(3.0, c[2], -c[6]);
    double uy = -2.0 * c[1] + fma(3.0, c[3], -c[7]);
    double vx = fma(-1.0, c[0], 3.0 * c[4]) + -2.0 * c[6];
    double vy = fma(-1.0, c[1], 3.0 * c[5]) + -2.0 * c[7];
    double uxx = ux * ux, uyy = uy * uy, vxx = vx * vx, vyy = vy * vy;
    double mu = uxx > uyy ? uxx : uyy;
    double mv = vxx > vyy ? vxx : vyy;
    return mu + mv;
}

// de Casteljau at t = 1/2 with the reference's 8x4 weight matrix (rows as dot4, shortened as described above):
//   l0 = (1, 0, 0, 0)              l1 = (.5, .5, 0, 0)   l2 = (.25, .5, .25, 0)   l3 = r0 = (.125, .375, .375, .125)
//   r1 = (0, .25, .5, .25)         r2 = (0, 0, .5, .5)   r3 = (0, 0, 0, 1)
SVGR_HD double split_mid(double x0, double x1, double x2, double x3) { return dot4(0.125, 0.375, 0.375, 0.125, x0, x1, x2, x3); }
SVGR_HD void cubic_left(const double* c, double* l) {
    for (int ax = 0; ax < 2; ++ax) {
        double x0 = c[ax], x1 = c[2 + ax], x2 = c[4 + ax], x3 = c[6 + ax];
        l[ax] = x0;
        l[2 + ax] = 0.5 * x0 + 0.5 * x1;
        l[4 + ax] = fma(0.25, x0, 0.25 * x2) + 0.5 * x1;
        l[6 + ax] = split_mid(x0, x1, x2, x3);
    }
}
SVGR_HD void cubic_right(const double* c, double* r) {
    for (int ax = 0; ax < 2; ++ax) {
        double x0 = c[ax], x1 = c[2 + ax], x2 = c[4 + ax], x3 = c[6 + ax];
        r[ax] = split_mid(x0, x1, x2, x3);
        r[2 + ax] = 0.5 * x2 + fma(0.25, x1, 0.25 * x3);
        r[4 + ax] = 0.5 * x2 + 0.5 * x3;
        r[6 + ax] = x3;
    }
}
// The two halves IN PLACE (same rows, evaluated in an order in which every input is still the old value): no
// temporary cubic and no copy back, which is a quarter of the instructions of the flatten kernel otherwise.
SVGR_HD void half_left(double& x0, double& x1, double& x2, double& x3) {
    x3 = split_mid(x0, x1, x2, x3);
    x2 = fma(0.25, x0, 0.25 * x2) + 0.5 * x1;
    x1 = 0.5 * x0 + 0.5 * x1;
}
SVGR_HD void half_right(double& x0, double& x1, double& x2, double& x3) {
    x0 = split_mid(x0, x1, x2, x3);
    x1 = 0.5 * x2 + fma(0.25, x1, 0.25 * x3);
    x2 = 0.5 * x2 + 0.5 * x3;
}
SVGR_HD void cubic_left_inplace(double* c) {
    half_left(c[0], c[2], c[4], c[6]);
    half_left(c[1], c[3], c[5], c[7]);
}
SVGR_HD void cubic_right_inplace(double* c) {
    half_right(c[0], c[2], c[4], c[6]);
    half_right(c[1], c[3], c[5], c[7]);
}
SVGR_HD void cubic_split(const double* c, double* l, double* r) {
    cubic_left(c, l);
    cubic_right(c, r);
}

// Depth-first adaptive subdivision. `emit(p0r, p0c, p1r, p1c)` is called once per flat piece,
// in curve order. Returns the number of pieces, or -1 when the depth cap was hit (non-finite
// or absurd input; the reference would never terminate there).
template <class Emit>
SVGR_HD int flatten_cubic(const double* cubic, double thr, Emit&& emit) {
    double stack[kMaxFlattenDepth][8];
    double cur[8];
    for (int i = 0; i < 8; ++i) cur[i] = cubic[i];
    int sp = 0, n = 0;
    bool overflow = false;
    for (;;) {
        bool flat = cubic_flatness(cur) < thr;
        if (!flat && sp >= kMaxFlattenDepth) { flat = true; overflow = true; }
        if (flat) {
            emit(cur[0], cur[1], cur[6], cur[7]);
            ++n;
            if (sp == 0) break;
            --sp;
            for (int i = 0; i < 8; ++i) cur[i] = stack[sp][i];
        } else {
            double l[8];
            cubic_split(cur, l, stack[sp]);
            ++sp;
            for (int i = 0; i < 8; ++i) cur[i] = l[i];
        }
    }
    return overflow ? -1 : n;
}

// Stack-free depth-first subdivision of the subtree under `root` (whose ancestors the caller has
// already found non-flat).  The position is a (level, path-bits) pair; stepping to a right sibling
// recomputes the node from `root` by `level` half-splits, which is the same sequence of roundings
// the recursive form performs, so the pieces are bit-identical -- and nothing spills to scratch.
// `emit(p0r, p0c, p1r, p1c)` once per flat piece in curve order; returns the piece count and sets
// `overflow` when `max_depth` forced a piece out (non-finite / absurd input).
// `ends` (optional): the end points of the first two pieces {r1, c1, r2, c2}, so that a caller that first counts and then
// stores can skip the second traversal for the (very common) subtrees of one or two pieces.
// (`NE`: how many first end points `ends` receives, {r, c} each)
template <int NE = 2, class Emit>
SVGR_HD int flatten_subtree(const double* root, double thr, int max_depth, Emit&& emit, bool& overflow, double* ends = nullptr) {
    double cur[8];
    for (int i = 0; i < 8; ++i) cur[i] = root[i];
    int level = 0, n = 0;
    unsigned long long idx = 0;
    double er[NE], ec[NE];
    for (int k = 0; k < NE; ++k) er[k] = ec[k] = 0.0;
    for (;;) {
        bool flat = cubic_flatness(cur) < thr;
        if (!flat && level >= max_depth) { flat = true; overflow = true; }
        if (flat) {
            emit(cur[0], cur[1], cur[6], cur[7]);
            for (int k = 0; k < NE; ++k) { er[k] = n == k ? cur[6] : er[k]; ec[k] = n == k ? cur[7] : ec[k]; }
            ++n;
            while (level > 0 && (idx & 1ull)) { idx >>= 1; --level; }
            if (level == 0) break;
            idx |= 1ull;
            for (int i = 0; i < 8; ++i) cur[i] = root[i];
            for (int l = level - 1; l >= 0; --l) {
                if ((idx >> l) & 1ull) cubic_right_inplace(cur); else cubic_left_inplace(cur);
            }
        } else {
            cubic_left_inplace(cur);
            ++level;
            idx <<= 1;
        }
    }
    if (ends)
        for (int k = 0; k < NE; ++k) { ends[2 * k] = er[k]; ends[2 * k + 1] = ec[k]; }
    return n;
}

// ------------------------------------------------------------------------------------
// order-preserving double <-> uint64 key (for integer atomicMin / atomicMax on coordinates)
// ------------------------------------------------------------------------------------
SVGR_HD uint64_t f64_key(double v) {
    union { double d; uint64_t u; } x;
    x.d = v;
    return (x.u >> 63) ? ~x.u : (x.u | 0x8000000000000000ull);
}
SVGR_HD double key_f64(uint64_t k) {
    union { double d; uint64_t u; } x;
    x.u = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return x.d;
}

// double -> int for values that are integral already (floor / ceil results) or get truncated toward zero.
// On the device v_cvt_i32_f64 saturates by itself (and gives 0 for a NaN); the host build clamps explicitly.
// Coordinates beyond +-1e9 are rejected before they get here (k_path_bbox), so both forms agree on everything used.
SVGR_HD int clamp_to_int(double v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __double2int_rz(v);
#else
    v = v < -1.0e9 ? -1.0e9 : (v > 1.0e9 ? 1.0e9 : v);
    return (int)v;
#endif
}

// ------------------------------------------------------------------------------------
// One edge of a path, prepared the way line_signed_coverage prepares it (S:2230-2242):
// coordinates relative to the layer origin, oriented so rows increase.
// ------------------------------------------------------------------------------------
struct EdgeSetup {
    double p0y, p1y, dxdy, x, dir;  // x = column at the first traced row's entry
    int y_begin, y_end;             // traced rows [y_begin, y_end) (clipped to [0, rows))
    bool valid;
};

SVGR_HD EdgeSetup edge_setup(double ar, double ac, double br, double bc, int rows) {
    EdgeSetup e;
    e.valid = false;
    e.p0y = e.p1y = e.dxdy = e.x = 0.0;
    e.dir = 1.0;
    e.y_begin = e.y_end = 0;
    if (ar == br) return e;  // horizontal: no signed coverage
    double p0y = ar, p0x = ac, p1y = br, p1x = bc;
    if (!(ar < br)) {
        e.dir = -1.0;
        p0y = br; p0x = bc; p1y = ar; p1x = ac;
    }
    e.p0y = p0y;
    e.p1y = p1y;
    e.dxdy = (p1x - p0x) / (p1y - p0y);
    e.x = p0x;
    e.y_begin = clamp_to_int(p0y > 0.0 ? p0y : 0.0);
    if (p0y < 0.0) e.x -= p0y * e.dxdy;
    int yend = clamp_to_int(ceil(p1y));
    e.y_end = yend < rows ? yend : rows;
    e.valid = e.y_begin < e.y_end;
    return e;
}

// Row recurrence (S:2243-2248). After step(y): x/x_next are the columns where the edge enters
// and leaves row y, d the signed height.
struct RowState {
    double x_next;  // carried between rows
    double x, d;
};

SVGR_HD void row_step(RowState& s, int y, double p0y, double p1y, double dxdy, double dir) {
    s.x = s.x_next;
    double yhi = (double)(y + 1) < p1y ? (double)(y + 1) : p1y;
    double ylo = (double)y > p0y ? (double)y : p0y;
    double dy = yhi - ylo;
    s.d = dir * dy;
    s.x_next = s.x + dxdy * dy;
}

// Area pieces of one row (S:2250-2303). `put(xi, v)` receives the UNCLAMPED column index; it
// returns false to stop the row (the reference `continue`s once a column is >= w).
template <class Put>
SVGR_HD void row_pieces(double x, double x_next, double d, Put&& put) {
    double x0 = x < x_next ? x : x_next;
    double x1 = x < x_next ? x_next : x;
    double x0_floor = floor(x0);
    int x0i = clamp_to_int(x0_floor);
    double x1_ceil = ceil(x1);
    int x1i = clamp_to_int(x1_ceil);
    if (x1i <= x0i + 1) {
        double xmf = 0.5 * (x + x_next) - x0_floor;
        if (!put(x0i, d * (1 - xmf))) return;
        put(x0i + 1, d * xmf);
    } else {
        double s = 1 / (x1 - x0);
        double x0f = x0 - x0_floor;
        double x1f = x1 - x1_ceil + 1.0;
        double o = 1 - x0f;
        double a0 = 0.5 * s * (o * o);
        double am = 0.5 * s * (x1f * x1f);
        if (!put(x0i, d * a0)) return;
        if (x1i == x0i + 2) {
            if (!put(x0i + 1, d * (1.0 - a0 - am))) return;
        } else {
            double a1 = s * (1.5 - x0f);
            if (!put(x0i + 1, d * (a1 - a0))) return;
            double ds = d * s;
            for (int xi = x0i + 2; xi < x1i - 1; ++xi)
                if (!put(xi, ds)) return;
            double a2 = a1 + (double)(x1i - x0i - 3) * s;
            if (!put(x1i - 1, d * (1.0 - a2 - am))) return;
        }
        put(x1i, d * am);
    }
}

// The same pieces in closed form, so that they can be computed once per edge row and applied by
// every tile that the row touches:
//   n == 1 : pieces at x0i (v[0]) and x0i+1 (v[1])                                  (one-pixel case)
//   n == 2 : x0i (v[0]), x0i+1 (v[1]), x0i+2 (v[4])
//   n >= 3 : x0i (v[0]), x0i+1 (v[1]), x0i+2 .. x0i+n-2 (v[2] each), x0i+n-1 (v[3]), x0i+n (v[4])
// where n = x1i - x0i (n = 1 also covers x1i == x0i).  Values are exactly those of row_pieces.
struct RowPieces {
    int x0i, n;
    double v[5];
};

SVGR_HD RowPieces row_record(double x, double x_next, double d) {
    RowPieces r;
    double x0 = x < x_next ? x : x_next;
    double x1 = x < x_next ? x_next : x;
    double x0_floor = floor(x0);
    r.x0i = clamp_to_int(x0_floor);
    double x1_ceil = ceil(x1);
    int x1i = clamp_to_int(x1_ceil);
    r.v[2] = r.v[3] = r.v[4] = 0.0;
    if (x1i <= r.x0i + 1) {
        double xmf = 0.5 * (x + x_next) - x0_floor;
        r.n = 1;
        r.v[0] = d * (1 - xmf);
        r.v[1] = d * xmf;
    } else {
        double s = 1 / (x1 - x0);
        double x0f = x0 - x0_floor;
        double x1f = x1 - x1_ceil + 1.0;
        double o = 1 - x0f;
        double a0 = 0.5 * s * (o * o);
        double am = 0.5 * s * (x1f * x1f);
        r.n = x1i - r.x0i;
        r.v[0] = d * a0;
        r.v[4] = d * am;
        if (r.n == 2) {
            r.v[1] = d * (1.0 - a0 - am);
        } else {
            double a1 = s * (1.5 - x0f);
            r.v[1] = d * (a1 - a0);
            r.v[2] = d * s;
            double a2 = a1 + (double)(x1i - r.x0i - 3) * s;
            r.v[3] = d * (1.0 - a2 - am);
        }
    }
    return r;
}

// k_path_build<2>: an UPPER BOUND of the adds the rows of one edge leave in ONE column tile of one band, from where the edge
// enters the band (`x_in`, the column at the start of its first row there) and where it leaves it (`x_out`), over `nr` rows of a
// layer `cols` wide whose tiles cut a run of equal pieces every `px` columns.  A row leaves two pieces (the pixel it lies in, the
// carry into the next) plus one per column border it crosses (row_record: n + 1 pieces, n = borders + 1); x runs monotonically along
// an edge, so over the rows that is 2 nr + |floor(x_out) - floor(x_in)| -- exactly; the closed-form ends are widened by a relative 1e-9
// so that a value the row recurrence rounds to the other side of an integer counts as crossing it.  Borders outside the layer make
// no piece of their own: right of it nothing is stored, left of it every piece folds into column 0 (up to five adds there per row
// instead of two).  A long span is cut into runs: four single pieces and a run piece per `px` columns -- 6 nr + borders / px + 1 --,
// 13 per row and tile at most (tiles are 64 columns, px = 8).  `lo` / `hi`: the first and last column the rows can touch (before the
// carry piece's + 1), for the caller's tile range.  tests/test_core_host.py::test_band_room_bounds_every_cell checks it against
// the pieces row_step / row_record / record_adds make, cell by cell, on millions of random edges.
SVGR_HD int band_room(double x_in, double x_out, int nr, int cols, int px, int& lo, int& hi) {
    double xlo = x_in < x_out ? x_in : x_out, xhi = x_in < x_out ? x_out : x_in;
    xlo -= 1e-9 * (1.0 + fabs(xlo));
    xhi += 1e-9 * (1.0 + fabs(xhi));
    lo = clamp_to_int(floor(xlo));
    hi = clamp_to_int(floor(xhi));
    const int lo_c = lo > -1 ? lo : -1;
    int hi_c = hi > -1 ? hi : -1;
    hi_c = hi_c < cols ? hi_c : cols;
    const int nc = hi_c - lo_c, fold = lo < 0 ? 3 * nr : 0;
    int room = 2 * nr + nc;
    room = room < 6 * nr + nc / px + 1 ? room : 6 * nr + nc / px + 1;
    return (room < 13 * nr ? room : 13 * nr) + fold;
}

// feed the pieces of a record to `put(xi, v)` in increasing column order; put returns false to stop
template <class Put>
SVGR_HD void apply_record(int x0i, int n, const double* v, Put&& put) {
    if (!put(x0i, v[0])) return;
    if (!put(x0i + 1, v[1])) return;
    if (n >= 3) {
        for (int xi = x0i + 2; xi < x0i + n - 1; ++xi)
            if (!put(xi, v[2])) return;
        if (!put(x0i + n - 1, v[3])) return;
    }
    if (n >= 2) put(x0i + n, v[4]);
}

// column span touched by a row, without computing the pieces: [lo, hi] inclusive
SVGR_HD void row_span(double x, double x_next, int& lo, int& hi) {
    double x0 = x < x_next ? x : x_next;
    double x1 = x < x_next ? x_next : x;
    lo = clamp_to_int(floor(x0));
    int x1i = clamp_to_int(ceil(x1));
    hi = (x1i <= lo + 1) ? lo + 1 : x1i;
}

// ------------------------------------------------------------------------------------
// fill rules + zero cut (S:984-990)
// ------------------------------------------------------------------------------------
// *_raw: before the zero cut (callers that only need "is it >= 1e-6" skip the select)
SVGR_HD double fill_nonzero_raw(double s) {
    double m = fabs(s);
    return m > 1.0 ? 1.0 : m;
}
SVGR_HD double fill_nonzero(double s) {
    double m = fill_nonzero_raw(s);
    return m < kZeroCut ? 0.0 : m;
}
SVGR_HD double fill_evenodd_raw(double s) {
    double a = s + 1.0;
    double r = a - 2.0 * floor(a * 0.5);
    return fabs(r - 1.0);
}
SVGR_HD double fill_evenodd(double s) {
    // np.remainder(a, 2.0) (sign follows the divisor) without fmod: a/2, floor, *2 and the final
    // subtraction are all exact for a power-of-two divisor, so r == a - 2*floor(a/2) bit for bit
    double a = s + 1.0;
    double r = a - 2.0 * floor(a * 0.5);
    double m = fabs(r - 1.0);
    return m < kZeroCut ? 0.0 : m;
}
SVGR_HD double fill_rule(double s, int rule) { return rule ? fill_evenodd(s) : fill_nonzero(s); }

// ------------------------------------------------------------------------------------
// source-over of one premultiplied pixel: dst = src + dst * (1 - src_a)   (S:286)
// ------------------------------------------------------------------------------------
SVGR_HD void over_px(double* dst, double s0, double s1, double s2, double s3) {
    double k = 1 - s3;
    dst[0] = s0 + dst[0] * k;
    dst[1] = s1 + dst[1] * k;
    dst[2] = s2 + dst[2] * k;
    dst[3] = s3 + dst[3] * k;
}

// float32-output variant: dst = fma(dst, 1 - src_a, src).  One rounding less than the reference's
// mul-then-add per channel (<= 0.5 ulp of a double closer to the exact value); only used where the
// canvas is rounded to float32 on store, the double outputs keep over_px.
SVGR_HD void over_px_fma(double* dst, double s0, double s1, double s2, double s3) {
    double k = 1 - s3;
    dst[0] = fma(dst[0], k, s0);
    dst[1] = fma(dst[1], k, s1);
    dst[2] = fma(dst[2], k, s2);
    dst[3] = fma(dst[3], k, s3);
}

}  // namespace svgr
