// svgr_stroke.cpp -- Path.stroke on the host: the reference's stroker restated in C++.
//
// The reference converts a stroked path into a fill path before anything is transformed or rasterised
// (Scene.render, S:668): offset every segment to both sides (line_offset S:2328-2337, bezier3_offset S:2113-2179,
// Tiller-Hanson on the control polygon with adaptive splitting), connect consecutive pieces with joins
// (stroke_line_join S:1495-1522), close the ends with caps (stroke_line_cap S:1466-1492) and walk back along the
// other side (Path.stroke S:1105-1180).  It is resolution independent, pure scalar double arithmetic and 0.3-0.5 s of
// Python per tiger render -- the Amdahl floor once rasterisation takes a millisecond -- hence native, hence here.
//
// Arithmetic follows the reference expression by expression (compiled with -ffp-contract=off; explicit fma only
// where numpy's matmul fuses).  Quadratic and arc segments are turned into cubics by the caller (geometry.py uses
// the same conversions as for fills), so the input here is lines and cubics only.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/svgr.h"

namespace {

constexpr double kEps = 2.220446049250313e-16;                      // sys.float_info.epsilon (S:40)
const double kCircle = 4 * (std::sqrt(2.0) - 1) / 3;               // CIRCLE_BEIZER_OFFSET (S:1463)

struct Pt {
    double x, y;
};
struct Curve {
    int n = 0;  // 2 line, 3 quad, 4 cubic
    Pt p[4];
    const Pt& back() const { return p[n - 1]; }
};
static Curve line_of(Pt a, Pt b) {
    Curve c;
    c.n = 2; c.p[0] = a; c.p[1] = b;
    return c;
}

// np.allclose(a, b): |a - b| <= 1e-8 + 1e-5 * |b| for every element
static bool allclose(Pt a, Pt b) {
    return std::fabs(a.x - b.x) <= 1e-8 + 1e-5 * std::fabs(b.x) && std::fabs(a.y - b.y) <= 1e-8 + 1e-5 * std::fabs(b.y);
}

// line_offset (S:2328-2337); false: the segment is too short to have a direction
static bool line_offset(Pt a, Pt b, double distance, Pt& o0, Pt& o1) {
    const double vx = b.x - a.x, vy = b.y - a.y;
    double len = vx * vx + vy * vy;
    if (len < kEps) return false;
    len = std::sqrt(len);
    const double dx = -vy * distance / len;
    const double dy = vx * distance / len;
    o0 = Pt{a.x + dx, a.y + dy};
    o1 = Pt{b.x + dx, b.y + dy};
    return true;
}

// line_intersect (S:2307-2325); false: parallel
static bool line_intersect(Pt a0, Pt a1, Pt b0, Pt b1, Pt& p, double& t0, double& t1) {
    const double x1 = a0.x, y1 = a0.y, x2 = a1.x, y2 = a1.y, x3 = b0.x, y3 = b0.y, x4 = b1.x, y4 = b1.y;
    const double det = (x4 - x3) * (y1 - y2) - (x1 - x2) * (y4 - y3);
    t0 = t1 = 0.0;
    if (std::fabs(det) < kEps) return false;
    t0 = ((y3 - y4) * (x1 - x3) + (x4 - x3) * (y1 - y3)) / det;
    t1 = ((y1 - y2) * (x1 - x3) + (x2 - x1) * (y1 - y3)) / det;
    p = Pt{x1 * (1 - t0) + x2 * t0, y1 * (1 - t0) + y2 * t0};
    return true;
}

// stroke_line_cap (S:1466-1492); cap: 0 butt, 1 round, 2 square
static void line_cap(Pt p0, Pt p1, int cap, std::vector<Curve>& out) {
    if (allclose(p0, p1)) return;
    if (cap == SVGR_CAP_BUTT) {
        out.push_back(line_of(p0, p1));
    } else if (cap == SVGR_CAP_ROUND) {
        Pt seg{p1.x - p0.x, p1.y - p0.y};
        const double radius = std::sqrt(seg.x * seg.x + seg.y * seg.y) / 2;
        seg.x /= 2 * radius; seg.y /= 2 * radius;
        const Pt nrm{-seg.y, seg.x};
        const double offset = kCircle * radius;
        const Pt center{(p0.x + p1.x) / 2, (p0.y + p1.y) / 2};
        const Pt mid{center.x + nrm.x * radius, center.y + nrm.y * radius};
        Curve a, b;
        a.n = b.n = 4;
        a.p[0] = p0;
        a.p[1] = Pt{p0.x + nrm.x * offset, p0.y + nrm.y * offset};
        a.p[2] = Pt{mid.x - seg.x * offset, mid.y - seg.y * offset};
        a.p[3] = mid;
        b.p[0] = mid;
        b.p[1] = Pt{mid.x + seg.x * offset, mid.y + seg.y * offset};
        b.p[2] = Pt{p1.x + nrm.x * offset, p1.y + nrm.y * offset};
        b.p[3] = p1;
        out.push_back(a);
        out.push_back(b);
    } else {  // square
        const Pt seg{p1.x - p0.x, p1.y - p0.y};
        const Pt nrm{-seg.y, seg.x};
        const Pt q0{p0.x + nrm.x / 2, p0.y + nrm.y / 2}, q1{p1.x + nrm.x / 2, p1.y + nrm.y / 2};
        out.push_back(line_of(p0, q0));
        out.push_back(line_of(q0, q1));
        out.push_back(line_of(q1, p1));
    }
}

// stroke_curve_tangent (S:1525-1535): first and last control-polygon leg that is not degenerate
static bool curve_tangent(const Curve& c, Pt* first, Pt* last) {
    bool any = false;
    for (int i = 0; i + 1 < c.n; ++i) {
        if (allclose(c.p[i], c.p[i + 1])) continue;
        if (!any) { first[0] = c.p[i]; first[1] = c.p[i + 1]; }
        last[0] = c.p[i]; last[1] = c.p[i + 1];
        any = true;
    }
    return any;
}

// stroke_line_join (S:1495-1522); join: 0 miter, 1 round, 2 bevel; miterlimit 4
static void line_join(const Curve& c0, const Curve& c1, int join, std::vector<Curve>& out) {
    const Pt a = c0.back(), b = c1.p[0];
    if (join == SVGR_JOIN_BEVEL) { out.push_back(line_of(a, b)); return; }
    Pt f0[2], l0[2], f1[2], l1[2];
    if (!curve_tangent(c0, f0, l0) || !curve_tangent(c1, f1, l1)) { out.push_back(line_of(a, b)); return; }
    if (allclose(l0[1], f1[0])) return;
    Pt p;
    double t0, t1;
    const bool hit = line_intersect(l0[0], l0[1], f1[0], f1[1], p, t0, t1);
    if (!hit || (0 <= t0 && t0 <= 1 && 0 <= t1 && t1 <= 1)) { out.push_back(line_of(a, b)); return; }
    if (std::fabs(t0) < 4 && std::fabs(t1) < 4) {
        if (join == SVGR_JOIN_MITER) {
            out.push_back(line_of(a, p));
            out.push_back(line_of(p, b));
        } else {
            Curve q;
            q.n = 3; q.p[0] = a; q.p[1] = p; q.p[2] = b;
            out.push_back(q);
        }
        return;
    }
    out.push_back(line_of(a, b));
}

// np.matmul(BEZIER3_SPLIT, points) (S:2058-2063): dgemm accumulates k = 0..3 with fused multiply-adds
static double mm4(double w0, double w1, double w2, double w3, double x0, double x1, double x2, double x3) {
    return std::fma(w3, x3, std::fma(w2, x2, std::fma(w1, x1, w0 * x0)));
}
static void split(const Curve& c, Curve& l, Curve& r) {
    l.n = r.n = 4;
    for (int ax = 0; ax < 2; ++ax) {
        auto get = [&](const Pt& q) { return ax ? q.y : q.x; };
        auto set = [&](Pt& q, double v) { (ax ? q.y : q.x) = v; };
        const double x0 = get(c.p[0]), x1 = get(c.p[1]), x2 = get(c.p[2]), x3 = get(c.p[3]);
        set(l.p[0], mm4(1, 0, 0, 0, x0, x1, x2, x3));
        set(l.p[1], mm4(0.5, 0.5, 0, 0, x0, x1, x2, x3));
        set(l.p[2], mm4(0.25, 0.5, 0.25, 0, x0, x1, x2, x3));
        set(l.p[3], mm4(0.125, 0.375, 0.375, 0.125, x0, x1, x2, x3));
        set(r.p[0], mm4(0.125, 0.375, 0.375, 0.125, x0, x1, x2, x3));
        set(r.p[1], mm4(0, 0.25, 0.5, 0.25, x0, x1, x2, x3));
        set(r.p[2], mm4(0, 0, 0.5, 0.5, x0, x1, x2, x3));
        set(r.p[3], mm4(0, 0, 0, 1, x0, x1, x2, x3));
    }
}

// should_split of bezier3_offset (S:2121-2137)
static bool should_split(const Curve& c) {
    const Pt c0 = c.p[0], c1 = c.p[1], c2 = c.p[2], c3 = c.p[3];
    const Pt base{c3.x - c0.x, c3.y - c0.y};
    const Pt mid{c2.x - c1.x, c2.y - c1.y};
    if (base.x * mid.x + base.y * mid.y < 0) return true;  // angle(c3 - c0, c2 - c1) beyond +-90 degrees
    const Pt d1{c1.x - c0.x, c1.y - c0.y}, d2{c2.x - c0.x, c2.y - c0.y};
    const double a0 = base.x * d1.y - base.y * d1.x, a1 = base.x * d2.y - base.y * d2.x;
    if (a0 * a1 < 0) return true;  // control points on different sides of the baseline
    const Pt mass{(((c0.x + c1.x) + c2.x) + c3.x) / 4, (((c0.y + c1.y) + c2.y) + c3.y) / 4};
    const Pt half{mm4(0.125, 0.375, 0.375, 0.125, c0.x, c1.x, c2.x, c3.x), mm4(0.125, 0.375, 0.375, 0.125, c0.y, c1.y, c2.y, c3.y)};
    const double ex = mass.x - half.x, ey = mass.y - half.y;
    const double dist = ex * ex + ey * ey;
    auto mx = [](double a, double b) { return a > b ? a : b; };
    auto mn = [](double a, double b) { return a < b ? a : b; };
    const double wx = mx(mx(c0.x, c1.x), mx(c2.x, c3.x)) - mn(mn(c0.x, c1.x), mn(c2.x, c3.x));
    const double wy = mx(mx(c0.y, c1.y), mx(c2.y, c3.y)) - mn(mn(c0.y, c1.y), mn(c2.y, c3.y));
    return dist * 100 > wx * wx + wy * wy;
}

// bezier3_offset (S:2113-2179)
// Returns false when the subdivision does not converge: a piece that keeps asking to be split while producing no
// offsetable line (e.g. a cubic whose first three points coincide) is halved for ever -- the reference spins on such a
// curve (S:2141-2145); here the stroke is refused after 65 536 pieces.
static bool cubic_offset(const Curve& cubic, double distance, std::vector<Curve>& outputs_all) {
    std::vector<Curve> outputs, stack{cubic};
    int budget = 1 << 16;
    while (!stack.empty()) {
        if (--budget < 0) return false;
        const Curve c = stack.back();
        stack.pop_back();
        // A piece whose consecutive points all coincide (to the tolerance below) has no offsetable line, and neither has
        // any part of it: it contributes nothing, however often the reference goes on halving it.  Dropping it here is
        // what makes a cubic with coincident leading control points terminate.
        if (allclose(c.p[0], c.p[1]) && allclose(c.p[1], c.p[2]) && allclose(c.p[2], c.p[3])) continue;
        if (should_split(c) && outputs.size() < 16) {
            Curve l, r;
            split(c, l, r);
            stack.push_back(r);
            stack.push_back(l);
            continue;
        }
        Curve out;
        out.n = 0;
        int repeat = 0;
        bool have_line = false;
        Pt line0{0, 0}, line1{0, 0}, o0{0, 0}, o1{0, 0};
        for (int i = 0; i + 1 < 4; ++i) {
            const Pt p0 = c.p[i], p1 = c.p[i + 1];
            if (allclose(p0, p1)) { ++repeat; continue; }  // not offsetable
            if (!line_offset(p0, p1, distance, o0, o1)) { ++repeat; continue; }  // (the reference would raise here)
            if (have_line) {
                Pt x;
                double t0, t1;
                if (line_intersect(line0, line1, o0, o1, x, t0, t1)) o0 = x;
                else o0 = Pt{(line1.x + o0.x) / 2, (line1.y + o0.y) / 2};
            }
            for (int k = 0; k < repeat + 1 && out.n < 4; ++k) out.p[out.n++] = o0;
            repeat = 0;
            line0 = o0; line1 = o1;
            have_line = true;
        }
        if (have_line) {
            for (int k = 0; k < repeat + 1 && out.n < 4; ++k) out.p[out.n++] = o1;
            if (!outputs.empty() && !allclose(out.p[0], outputs.back().back()))
                line_cap(out.p[0], outputs.back().back(), SVGR_CAP_ROUND, outputs);  // "M0,0 C100,50 0,50 100,0" (S:2171)
            outputs.push_back(out);
        }
    }
    outputs_all.insert(outputs_all.end(), outputs.begin(), outputs.end());
    return true;
}

static Curve reversed(const Curve& c) {
    Curve r;
    r.n = c.n;
    for (int i = 0; i < c.n; ++i) r.p[i] = c.p[c.n - 1 - i];
    return r;
}

}  // namespace

struct svgr_stroke_out {
    std::vector<int32_t> types;
    std::vector<double> params;  // 8 per segment
    std::vector<int32_t> sizes;
    void add_subpath(const std::vector<Curve>& curves) {
        for (const Curve& c : curves) {
            types.push_back(c.n == 2 ? SVGR_PATH_LINE : (c.n == 3 ? SVGR_PATH_QUAD : SVGR_PATH_CUBIC));
            double q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int i = 0; i < c.n; ++i) { q[2 * i] = c.p[i].x; q[2 * i + 1] = c.p[i].y; }
            params.insert(params.end(), q, q + 8);
        }
        sizes.push_back((int32_t)curves.size());
    }
};

extern "C" {

static int path_stroke_impl(const int32_t* seg_types, const double* seg_params, const int32_t* subpath_sizes, int64_t n_subpaths,
                            double width, int linecap, int linejoin, svgr_stroke_out** out);

// No exception crosses the C ABI: running out of memory is a status like any other.
int svgr_path_stroke(const int32_t* seg_types, const double* seg_params, const int32_t* subpath_sizes, int64_t n_subpaths,
                     double width, int linecap, int linejoin, svgr_stroke_out** out) {
    try {
        return path_stroke_impl(seg_types, seg_params, subpath_sizes, n_subpaths, width, linecap, linejoin, out);
    } catch (const std::bad_alloc&) {
        return SVGR_E_NOMEM;
    } catch (...) {
        return SVGR_E_INVALID;
    }
}

static int path_stroke_impl(const int32_t* seg_types, const double* seg_params, const int32_t* subpath_sizes, int64_t n_subpaths,
                            double width, int linecap, int linejoin, svgr_stroke_out** out) {
    if (!out || n_subpaths < 0 || (n_subpaths > 0 && (!seg_types || !seg_params || !subpath_sizes))) return SVGR_E_INVALID;
    if (linecap < SVGR_CAP_BUTT || linecap > SVGR_CAP_SQUARE || linejoin < SVGR_JOIN_MITER || linejoin > SVGR_JOIN_BEVEL) return SVGR_E_INVALID;
    svgr_stroke_out* res = new (std::nothrow) svgr_stroke_out();
    if (!res) return SVGR_E_NOMEM;
    const double dist = width / 2;
    int64_t k = 0;
    for (int64_t s = 0; s < n_subpaths; ++s) {
        const int n = subpath_sizes[s];
        if (n <= 0) continue;
        std::vector<Curve> forward, backward;
        int last_type = -1;
        for (int i = 0; i < n; ++i, ++k) {
            const int t = seg_types[k];
            const double* q = seg_params + 8 * k;
            last_type = t;
            if (t == SVGR_PATH_LINE || t == SVGR_PATH_CLOSED) {
                Pt f0, f1, b0, b1;
                const Pt a{q[0], q[1]}, b{q[2], q[3]};
                if (!line_offset(a, b, dist, f0, f1)) continue;
                line_offset(a, b, -dist, b0, b1);
                forward.push_back(line_of(f0, f1));
                backward.push_back(line_of(b0, b1));
            } else if (t == SVGR_PATH_CUBIC) {
                Curve c;
                c.n = 4;
                for (int j = 0; j < 4; ++j) c.p[j] = Pt{q[2 * j], q[2 * j + 1]};
                if (!cubic_offset(c, dist, forward) || !cubic_offset(c, -dist, backward)) {
                    delete res;
                    return SVGR_E_OVERFLOW;  // the offset of this cubic does not converge (degenerate control points)
                }
            } else if (t == SVGR_PATH_UNCLOSED) {
                continue;
            } else {
                delete res;
                return SVGR_E_INVALID;  // quads / arcs are converted by the caller (S:1133-1140)
            }
        }
        const bool closed = last_type == SVGR_PATH_CLOSED;
        if (forward.empty()) continue;
        std::vector<Curve> curves;
        for (const Curve& c : forward) {
            if (!curves.empty()) line_join(curves.back(), c, linejoin, curves);
            curves.push_back(c);
        }
        if (closed) {
            line_join(curves.back(), curves.front(), linejoin, curves);
            res->add_subpath(curves);
            curves.clear();
        } else {
            line_cap(curves.back().back(), backward.back().back(), linecap, curves);
        }
        while (!backward.empty()) {
            const Curve c = reversed(backward.back());
            backward.pop_back();
            if (!curves.empty()) line_join(curves.back(), c, linejoin, curves);
            curves.push_back(c);
        }
        if (closed) line_join(curves.back(), curves.front(), linejoin, curves);
        else line_cap(curves.back().back(), curves.front().p[0], linecap, curves);
        res->add_subpath(curves);
    }
    *out = res;
    return SVGR_OK;
}

int svgr_stroke_out_counts(const svgr_stroke_out* s, int64_t* n_segs, int64_t* n_subpaths) {
    if (!s) return SVGR_E_INVALID;
    if (n_segs) *n_segs = (int64_t)s->types.size();
    if (n_subpaths) *n_subpaths = (int64_t)s->sizes.size();
    return SVGR_OK;
}

int svgr_stroke_out_copy(const svgr_stroke_out* s, int32_t* seg_types, double* seg_params, int32_t* subpath_sizes) {
    if (!s) return SVGR_E_INVALID;
    if (seg_types && !s->types.empty()) memcpy(seg_types, s->types.data(), sizeof(int32_t) * s->types.size());
    if (seg_params && !s->params.empty()) memcpy(seg_params, s->params.data(), sizeof(double) * s->params.size());
    if (subpath_sizes && !s->sizes.empty()) memcpy(subpath_sizes, s->sizes.data(), sizeof(int32_t) * s->sizes.size());
    return SVGR_OK;
}

void svgr_stroke_out_free(svgr_stroke_out* s) { delete s; }

}  // extern "C"
